/*
 * ttx.h -- C ABI of libttx.so: MI355X (gfx950) native TT-compressed EmbeddingBag.
 *
 * This header is the drop-in boundary for the hot path of
 * facebookresearch/FBTT-Embedding.  Every entry point below replaces one of the
 * eleven functions the reference exports from its native module `tt_embeddings`
 * (reference tt_embeddings.cpp:131-161); the reference prototype each one
 * replaces is cited next to it.  Signatures carry only plain pointers, sizes and
 * scalars (no torch / ATen types): the Python shim
 * fbtt-embedding_amd/tt_embeddings.py binds them with ctypes, and INTEGRATION.md
 * shows the pybind11 stub a maintainer of the reference would write instead.
 *
 * Conventions
 *  - All data pointers are DEVICE pointers (HBM) unless the name ends in _host.
 *    `tt_cores`, `optimizer_state`, `d_tt_cores` are HOST arrays of T device
 *    pointers (one per TT core).
 *  - Index tensors are int64 (as in the reference), cores/outputs fp32, all
 *    contiguous.  Core t has shape [num_tables, p[t], r[t]*q[t]*r[t+1]]; each
 *    p-slice is row-major [r[t]][q[t]][r[t+1]] (reference tt_embeddings_ops.py
 *    :513-530, :601-611).
 *  - `stream` is a hipStream_t.  All work is enqueued on it; nothing
 *    synchronises with the host except ttx_preprocess_indices_sync in the
 *    cache-live case (exactly like the reference, tt_embeddings_cuda.cu:1481-1488)
 *    and ttx_cache_populate (one 8-byte read-back to size the radix sort).
 *  - `workspace` is caller-provided device scratch of at least the size the
 *    matching *_workspace_bytes() query returns (256-byte aligned).  No entry
 *    point allocates device memory.
 *  - Return value: 0 on success, negative TTX_E* on error;
 *    ttx_last_error() returns a thread-local message for the last failure.
 *  - Strides L[t] = prod_{s>t} p[s] (reference tt_embeddings_ops.py:506-512) are
 *    derived from `p` on the host; the reference's device tensor `L` is not read.
 */
#ifndef TTX_H_
#define TTX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTX_MAX_CORES 4

#define TTX_OK 0
#define TTX_EINVAL (-1)    /* bad argument (mirrors the reference's TORCH_CHECKs) */
#define TTX_EWORKSPACE (-2) /* workspace too small */
#define TTX_EUNSUPPORTED (-3) /* geometry does not fit this build's LDS tiling */
#define TTX_EHIP (-4)       /* a HIP runtime call failed */

/* fused-optimizer selector of ttx_tt_backward (reference OPTIM_* enum,
 * tt_embeddings_cuda.cu:31-35) */
#define TTX_OPTIM_SGD 0
#define TTX_OPTIM_ADAGRAD 1
#define TTX_OPTIM_DENSE 2

typedef void* ttx_stream_t; /* hipStream_t */

#define TTX_MAX_TABLES_MIXED 64

/* TT geometry shared by all tables of one TableBatchedTTEmbeddingBag
 * (reference tt_embeddings_ops.py:459-488).
 *
 * p_tables (beyond the reference, which batches tables of ONE shape only,
 * tt_embeddings_ops.py:424): NULL, or a host array [num_tables][T] of per-table
 * row factors -- tables of different cardinality in one batched lookup
 * (num_tables <= TTX_MAX_TABLES_MIXED).  q and the ranks stay common, so every core
 * slice has one size and core t is ONE array of sum_k p_tables[k][t] slices, table
 * after table: tt_cores[t] points at [sum_k p_k_t][r_t q_t r_{t+1}] floats and the
 * slice of (table k, i_t) is base_t[k] + i_t with base_t[k] = sum_{j<k} p_tables[j][t].
 * p[] is ignored then.  Only the calls that take tableidx accept such a geometry (plan,
 * forward, backward; not the one-table cache entry points).  The array is read during
 * the call only. */
typedef struct ttx_geom {
  int32_t T;                    /* number of TT cores, 2..4 */
  int32_t num_tables;           /* tt_cores[t].size(0) */
  int32_t p[TTX_MAX_CORES];     /* tt_p_shapes */
  int32_t q[TTX_MAX_CORES];     /* tt_q_shapes */
  int32_t r[TTX_MAX_CORES + 1]; /* padded ranks [1, r1, .., 1] */
  const int32_t* p_tables;      /* NULL: every table has the row factors p[] */
} ttx_geom;

const char* ttx_last_error(void);
int ttx_version(void);

/* ------------------------------------------------------------------ plan ---
 * The lookup plan is this library's replacement for the reference's per-chunk
 * pointer-array set-up kernels (init_batch_gemm_{forward,backward}_*T_kernel,
 * tt_embeddings_cuda.cu:79-360, :754-918): index decode, a stable radix sort of
 * the lookups by (table, i_t) for every core and the work-list of index groups
 * that share a middle-core slice.  It depends only on (geometry, indices,
 * tableidx, rowidx); forward and backward of the same batch can share one plan.
 * rowidx may be NULL (the backward kernel then gathers the bag row per lookup).
 * Passing plan == NULL to ttx_tt_forward / ttx_tt_backward builds it inside
 * their workspace.
 * With FOUR cores on the three-core kernels a plan also carries the product of each
 * lookup's last two core slices, left there by the latest ttx_tt_forward on that plan;
 * a ttx_tt_backward on the same plan, with cores 2 and 3 at the same addresses, reads it
 * instead of recomputing it.  This library's fused optimizers and ttx_plan_build drop it;
 * a caller who rewrites cores 2 / 3 by other means BETWEEN the forward and the backward
 * of one plan (nothing in the reference's flow does) must rebuild the plan first. */
size_t ttx_plan_bytes(const ttx_geom* g, int64_t nnz);
int ttx_plan_build(const ttx_geom* g, int64_t nnz, const int64_t* indices,
                   const int64_t* tableidx, const int64_t* rowidx, void* plan,
                   size_t plan_bytes, ttx_stream_t stream);

/* --------------------------------------------------------------- forward ---
 * replaces tt_embeddings_forward_cuda (tt_embeddings.cpp:13-26,
 * tt_embeddings_cuda.cu:964-1075).  output[num_tables,B,D] is overwritten with
 * the bag sums of the first `nnz` lookups (zeros when nnz == 0).  D may be any
 * positive value (the reference requires D % 4 == 0). */
size_t ttx_tt_forward_workspace_bytes(const ttx_geom* g, int32_t B, int32_t D,
                                      int64_t nnz);
int ttx_tt_forward(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz,
                   const int64_t* indices, const int64_t* rowidx,
                   const int64_t* tableidx, const float* const* tt_cores,
                   float* output, const void* plan, void* workspace,
                   size_t workspace_bytes, ttx_stream_t stream);

/* nn.EmbeddingBag's per_sample_weights (not in the reference): lookup n enters its
 * bag scaled by per_sample_weights[n], and its share of the bag gradient is scaled the same way in
 * ttx_tt_backward_w.  NULL = the plain entry point. */
int ttx_tt_forward_w(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz,
                     const int64_t* indices, const int64_t* rowidx, const int64_t* tableidx,
                     const float* per_sample_weights, const float* const* tt_cores,
                     float* output, const void* plan, void* workspace, size_t workspace_bytes,
                     ttx_stream_t stream);

/* ... and with a gradient for the weights: ttx_tt_forward_wr leaves the lookups' rows (unweighted, [nnz, D], 16-byte
 * aligned) in rows_keep (NULL = ttx_tt_forward_w), and ttx_psw_backward turns them into
 * d_psw[n] = <d_output[tableidx[n], rowidx[n], :], rows[n, :]>, what autograd gives nn.EmbeddingBag(mode="sum"). */
int ttx_tt_forward_wr(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz,
                      const int64_t* indices, const int64_t* rowidx, const int64_t* tableidx,
                      const float* per_sample_weights, const float* const* tt_cores,
                      float* output, float* rows_keep, const void* plan, void* workspace,
                      size_t workspace_bytes, ttx_stream_t stream);

/* Not in the reference (round 4): the cache-live forward of ONE table in one call -- tt_embeddings_forward_cuda on the misses
 * followed by cache_forward_cuda (tt_embeddings_cuda.cu:1498-1572) on the hits -- with the bag sums of both parts in ONE launch.
 * The batch is the partitioned one of ttx_preprocess_indices_async: `nnz` lookups, the misses in front; `plan` is the plan of the
 * misses (ttx_plan_build_n with the device-side split point), `rowidx` / `cache_loc` are the partitioned bag rows / cache
 * locations of all `nnz` positions.  A bag's row receives at most two terms (its contraction sum, its cache sum) through fp32
 * atomics on the zeroed output: the result does not depend on their order.  ttx_tt_forward_cached_supported says whether the
 * call is taken this way (one table, D % 4 == 0, at most 65536 lookups); otherwise run ttx_tt_forward + ttx_cache_forward_n. */
int ttx_tt_forward_cached_supported(const ttx_geom* g, int32_t D, int64_t nnz);
int ttx_tt_forward_cached(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz,
                          const int64_t* indices, const int64_t* rowidx, const int64_t* tableidx,
                          const float* const* tt_cores, const int32_t* cache_loc, const float* cache_weight,
                          float* output, const void* plan, void* workspace, size_t workspace_bytes,
                          ttx_stream_t stream);

/* Not in the reference: ttx_tt_forward_wr with the bag pooling done INSIDE the contraction kernel (no pooling launch:
 * the lookup that completes a bag sums its rows in index order, same result bit for bit) where the shape and the
 * batch allow it -- the call falls back to the separate pooling launch otherwise, so it is always valid.
 *   offsets  [num_tables * B + 1] int64: the bags' extents in the table-major, include_last_offset form the rowidx /
 *            tableidx arrays were derived from (tt_embeddings_ops.py:851; every lookup of bag b at [offsets[b], offsets[b+1]))
 *   arrive   device array of ttx_tt_forward_arrive_ints() int32, ALL ZERO on entry; all zero again when the call's
 *            kernels have run (keep one per stream, zero it once).  NULL / 0 ints: pooling stays a launch of its own. */
int64_t ttx_tt_forward_arrive_ints(const ttx_geom* g, int64_t nnz);
int ttx_tt_forward_o(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* indices,
                     const int64_t* rowidx, const int64_t* tableidx, const float* per_sample_weights,
                     const float* const* tt_cores, float* output, float* rows_keep, const int64_t* offsets,
                     int32_t* arrive, const void* plan, void* workspace, size_t workspace_bytes, ttx_stream_t stream);
int ttx_psw_backward(int32_t B, int32_t D, int64_t nnz, const float* rows, const int64_t* rowidx,
                     const int64_t* tableidx, const float* d_output, float* d_psw, ttx_stream_t stream);

/* decompress rows: rows[n, :] = TT row of indices[n] in table tableidx[n]
 * (tableidx == NULL -> table 0).  This is the contraction alone, the part of
 * prefetch_cached_weights_cuda (tt_embeddings_cuda.cu:1156-1258) that fills
 * cache_weight.  Same workspace size as ttx_tt_forward with B = 0. */
int ttx_tt_rows(const ttx_geom* g, int32_t D, int64_t nnz,
                const int64_t* indices, const int64_t* tableidx,
                const float* const* tt_cores, float* rows, void* workspace,
                size_t workspace_bytes, ttx_stream_t stream);

/* -------------------------------------------------------------- backward ---
 * replaces tt_embeddings_backward_{dense,sgd,adagrad}_cuda
 * (tt_embeddings.cpp:28-72, tt_embeddings_cuda.cu:419-752).
 *  optim == TTX_OPTIM_DENSE  : d_tt_cores[t] (full core shape) is overwritten
 *                              with the gradient; cores untouched.
 *  optim == TTX_OPTIM_SGD    : cores  -= lr * g              (every element)
 *  optim == TTX_OPTIM_ADAGRAD: state += g*g;
 *                              cores  -= lr * g / (sqrtf(state) + eps)
 * Gradients of duplicate lookups are summed before the single optimizer step.
 * Unlike the reference's apply kernels (tt_embeddings_cuda.cu:612-648, whose
 * grid covers only part of each core when p[t] > r*q*r) every touched slice is
 * updated; slices with no lookup have g == 0 and are left unchanged. */
size_t ttx_tt_backward_workspace_bytes(const ttx_geom* g, int32_t B, int32_t D,
                                       int64_t nnz);
int ttx_tt_backward(const ttx_geom* g, int32_t optim, int32_t B, int32_t D,
                    float learning_rate, float eps, int64_t nnz,
                    const int64_t* indices, const int64_t* rowidx,
                    const int64_t* tableidx, const float* d_output,
                    float* const* tt_cores, float* const* optimizer_state,
                    float* const* d_tt_cores, const void* plan, void* workspace,
                    size_t workspace_bytes, ttx_stream_t stream);

int ttx_tt_backward_w(const ttx_geom* g, int32_t optim, int32_t B, int32_t D,
                      float learning_rate, float eps, int64_t nnz, const int64_t* indices,
                      const int64_t* rowidx, const int64_t* tableidx,
                      const float* per_sample_weights, const float* d_output,
                      float* const* tt_cores, float* const* optimizer_state,
                      float* const* d_tt_cores, const void* plan, void* workspace,
                      size_t workspace_bytes, ttx_stream_t stream);

/* Not in the reference (round 4): ttx_tt_backward_w of a cache-live batch with the cache rows' scatter
 * (cache_backward_sgd_cuda / _dense, tt_embeddings_cuda.cu:1574-1697: cache_dst[loc[n], :] += cache_scale * cache_grad[rowidx[n], :]
 * for the lookups behind the device-side split point *skip_dev) done by work-groups of the optimizer's launch instead of a launch of
 * its own.  *tail_done = 1 when it was; 0 (four-core route, an empty batch, switched off) = the caller runs
 * ttx_cache_backward_sgd_n / _dense_n itself.  rowidx is the partitioned bag rows of all nnz positions, as for ttx_tt_forward_cached. */
int ttx_tt_backward_wc(const ttx_geom* g, int32_t optim, int32_t B, int32_t D, float lr, float eps,
                       int64_t nnz, const int64_t* indices, const int64_t* rowidx,
                       const int64_t* tableidx, const float* per_sample_weights, const float* d_output,
                       float* const* tt_cores, float* const* optimizer_state, float* const* d_tt_cores,
                       const void* plan, void* workspace, size_t workspace_bytes, ttx_stream_t stream,
                       const int32_t* skip_dev, const int32_t* cache_loc, const float* cache_grad, float cache_scale,
                       float* cache_dst, int32_t* tail_done);

/* ----------------------------------------------- duplicate lookups -----
 * Not in the reference (which contracts every lookup on its own): a batch's lookups are mapped onto their
 * DISTINCT (table, index) pairs, the contraction runs once per pair, bag pooling gathers each lookup's row
 * through the map (same sums in the same order: the output is bit-identical to ttx_tt_forward's), and the
 * backward first adds up, in a fixed order, the bag gradients of a pair's occurrences -- one contraction, one
 * set of partial gradients per pair.  Pays on large batches that repeat rows (Zipf 1.2, 327k lookups: 0.79 ->
 * 0.35 ms/step); on a uniform stream it only adds the key sort and the launches of the map.
 *
 *   ttx_dedup_bytes     size of the map buffer; 0 = this batch is not deduplicated (per-table row factors, a
 *                       key space num_tables * prod(p) beyond 2^61, nnz >= 2^31): use the plain entry points.
 *   ttx_dedup_build     builds the map (<= 16384 lookups with 32-bit keys: one work-group sorts the keys in LDS;
 *                       otherwise a multi-work-group stable radix sort of 64-bit keys -- deterministic either
 *                       way, nothing read back: capturable) AND the lookup plan of the distinct pairs into
 *                       `plan` (ttx_plan_bytes(g, nnz) bytes).
 *   ttx_tt_forward_dd / ttx_tt_backward_dd   as ttx_tt_forward_w / ttx_tt_backward_w (psw may be NULL), driven
 *                       by a map + plan built for the same (indices, tableidx). */
size_t ttx_dedup_bytes(const ttx_geom* g, int64_t nnz);
int ttx_dedup_build(const ttx_geom* g, int64_t nnz, const int64_t* indices, const int64_t* tableidx,
                    void* dedup, size_t dedup_bytes, void* plan, size_t plan_bytes, ttx_stream_t stream);
size_t ttx_tt_forward_dd_workspace_bytes(const ttx_geom* g, int32_t D, int64_t nnz);
int ttx_tt_forward_dd(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* rowidx,
                      const int64_t* tableidx, const float* per_sample_weights, const void* dedup,
                      const void* plan, const float* const* tt_cores, float* output, void* workspace,
                      size_t workspace_bytes, ttx_stream_t stream);
size_t ttx_tt_backward_dd_workspace_bytes(const ttx_geom* g, int32_t D, int64_t nnz);
int ttx_tt_backward_dd(const ttx_geom* g, int32_t optim, int32_t B, int32_t D, float learning_rate, float eps,
                       int64_t nnz, const int64_t* rowidx, const int64_t* tableidx,
                       const float* per_sample_weights, const float* d_output, const void* dedup,
                       const void* plan, float* const* tt_cores, float* const* optimizer_state,
                       float* const* d_tt_cores, void* workspace, size_t workspace_bytes, ttx_stream_t stream);

/* ------------------------------------------- core-0 row split (not in the reference) -----
 * A T = 3 table whose first factor is q0 = k q0' (q0 = 8: k = 2, q0' = 4) is contracted as k PART lookups per index
 * in the table p' = [k p0, p1, p2], q' = [q0', q1, q2]: core 0 [p0, q0, r1] is, element for element, [k p0, q0', r1]; index
 * (i0, i1, i2) becomes (k i0 + h, i1, i2), h = 0..k-1, and part h of bag b is bag k b + h of a batch with k nb bags and
 * D / k columns -- an output [nb, D] row-major IS [k nb, D / k], so nothing is copied back.  This is how factorings with
 * q0 > 4 (the reference's default for D = 512, [8,8,8]) reach the shape-specialised kernels (q0 <= 4): same sums, the
 * core-1 / core-2 gradients of the parts add up in the ordinary reduction.  `offsets` holds nb + 1 entries (closing entry
 * included), bags table-major as everywhere; out_indices [k nnz], out_offsets [k nb + 1].  p_rest = p1 * p2. */
int ttx_split0_expand(int64_t nnz, int64_t nb, int32_t k, int64_t p_rest, const int64_t* indices, const int64_t* offsets,
                      int64_t* out_indices, int64_t* out_offsets, ttx_stream_t stream);

/* ------------------------------------------------------ software cache -----
 * replaces update_cache_state_cuda (tt_embeddings.cpp:74,
 * tt_embeddings_cuda.cu:1077-1113): cache_freq[slot(idx)] += 1 with at most 3
 * linear probes of an open-addressing table keyed by a murmur3-style hash
 * (hashtbl_cuda_utils.cuh:48-76, :102-133). */
int ttx_update_cache_state(int64_t nnz, const int64_t* indices,
                           int64_t hashtbl_size, int64_t* hashtbl,
                           int64_t* cache_freq, ttx_stream_t stream);

/* replaces preprocess_indices_sync_cuda (tt_embeddings.cpp:88-95,
 * tt_embeddings_cuda.cu:1377-1496).
 *  rowidx/tableidx[nnz] are always written (bag b covers
 *  [offsets[b], offsets[b+1]); rowidx = b % B, tableidx = b / B).
 *  If warmup != 0 or num_tables != 1 nothing else happens, *num_tt_host = nnz
 *  and *partitioned_host = 0.  Otherwise every index is looked up in the
 *  hash table; lookups whose slot has a cache row go to the REAR of
 *  part_colidx / part_rowidx / part_cache_locations in reverse order, the rest
 *  keep their order at the front (cub::DevicePartition::Flagged semantics), the
 *  count of front entries is copied to *num_tt_host after a stream
 *  synchronise, and *partitioned_host = 1.  part_cache_locations of front
 *  entries is -1 (uninitialised in the reference). */
size_t ttx_preprocess_workspace_bytes(int64_t nnz);
int ttx_preprocess_indices_sync(int64_t nnz, const int64_t* colidx,
                                int64_t num_bags_total, const int64_t* offsets,
                                int32_t num_tables, int32_t warmup,
                                int64_t hashtbl_size, const int64_t* hashtbl,
                                const int32_t* cache_state, int64_t* rowidx,
                                int64_t* tableidx, int64_t* part_colidx,
                                int64_t* part_rowidx,
                                int32_t* part_cache_locations,
                                int32_t* num_tt_host, int32_t* partitioned_host,
                                void* workspace, size_t workspace_bytes,
                                ttx_stream_t stream);

/* The same, with update_cache_state folded into the first kernel when upd_hashtbl /
 * upd_cache_freq are non-NULL (one launch instead of two; the order "count the batch's
 * indices, then look them up" of tt_embeddings_ops.py:827-846 is kept). */
int ttx_preprocess_indices_sync_fused(int64_t nnz, const int64_t* colidx,
                                      int64_t num_bags_total, const int64_t* offsets,
                                      int32_t num_tables, int32_t warmup,
                                      int64_t hashtbl_size, const int64_t* hashtbl,
                                      const int32_t* cache_state, int64_t* rowidx,
                                      int64_t* tableidx, int64_t* part_colidx,
                                      int64_t* part_rowidx, int32_t* part_cache_locations,
                                      int32_t* num_tt_host, int32_t* partitioned_host,
                                      int64_t* upd_hashtbl, int64_t* upd_cache_freq,
                                      void* workspace, size_t workspace_bytes,
                                      ttx_stream_t stream);

/* ---- device-side counts: the cache-live path without its host read-back -------------------
 * The reference copies the partition's split point to the host and synchronises
 * (tt_embeddings_cuda.cu:1481-1488) because its launch grids depend on it.  Here every kernel
 * can take its lookup count from device memory instead, with the host-side `nnz` an upper bound
 * that only sizes grids and workspaces:
 *  - ttx_preprocess_indices_async: as ttx_preprocess_indices_sync_fused, but when num_tt_dev
 *    (a device int32) is non-NULL the count of TT entries is written there and nothing is read
 *    back (*num_tt_host keeps nnz);
 *  - ttx_plan_build_n: nnz_dev (device int32, <= nnz) is the number of leading entries of
 *    indices / tableidx / rowidx to plan; a plan built this way may be handed to
 *    ttx_tt_forward / ttx_tt_backward together with the same upper bound nnz (their kernels
 *    are driven by the plan; workspaces are sized for nnz);
 *  - ttx_cache_*_n: skip_dev (device int32) = number of leading entries of cache_locations /
 *    rowidx that are NOT cached; the kernels work on [*skip_dev, nnz).
 * Passing NULL for the device pointer gives exactly the plain entry point. */
int ttx_preprocess_indices_async(int64_t nnz, const int64_t* colidx, int64_t num_bags_total,
                                 const int64_t* offsets, int32_t num_tables, int32_t warmup,
                                 int64_t hashtbl_size, const int64_t* hashtbl,
                                 const int32_t* cache_state, int64_t* rowidx, int64_t* tableidx,
                                 int64_t* part_colidx, int64_t* part_rowidx,
                                 int32_t* part_cache_locations, int32_t* num_tt_host,
                                 int32_t* partitioned_host, int32_t* num_tt_dev,
                                 int64_t* upd_hashtbl, int64_t* upd_cache_freq, void* workspace,
                                 size_t workspace_bytes, ttx_stream_t stream);
int ttx_plan_build_n(const ttx_geom* g, int64_t nnz, const int32_t* nnz_dev, const int64_t* indices,
                     const int64_t* tableidx, const int64_t* rowidx, void* plan, size_t plan_bytes,
                     ttx_stream_t stream);
int ttx_cache_forward_n(int32_t B, int64_t nnz, const int32_t* skip_dev, const int32_t* cache_locations,
                        const int64_t* rowidx, int32_t D, const float* cache_weight, float* output,
                        ttx_stream_t stream);
int ttx_cache_backward_sgd_n(int64_t nnz, const int32_t* skip_dev, int32_t D, const float* grad_output,
                             const int32_t* cache_locations, const int64_t* rowidx, float learning_rate,
                             float* cache_weight, ttx_stream_t stream);
int ttx_cache_backward_dense_n(int64_t nnz, const int32_t* skip_dev, int32_t D, const float* grad_output,
                               const int32_t* cache_locations, const int64_t* rowidx, int64_t cache_size,
                               float* grad_cache_weight, ttx_stream_t stream);
int ttx_cache_backward_rowwise_adagrad_approx_n(int64_t nnz, const int32_t* skip_dev, int32_t D,
                                                const float* grad_output, const int32_t* cache_locations,
                                                const int64_t* rowidx, float learning_rate, float eps,
                                                float* cache_optimizer_state, float* cache_weight,
                                                ttx_stream_t stream);

/* nn.EmbeddingBag's per_sample_weights with a LIVE cache (not in the reference, which has no weights at all):
 *  - ttx_preprocess_indices_async_w: as ttx_preprocess_indices_async; the weights travel with their lookups into
 *    partitioned_weights and every partitioned entry's original position goes to partitioned_origin (either may be
 *    NULL), so that the gradient of the weights can be handed back in the caller's order;
 *  - ttx_cache_forward_nw: output[rowidx[n]] += per_sample_weights[n] * cache_weight[loc[n]] for the cached entries
 *    (per_sample_weights == NULL: ttx_cache_forward_n);
 *  - ttx_cache_rows_n: rows[n] = cache_weight[loc[n]] for the cached entries -- with the contraction's rows of the
 *    misses in front (ttx_tt_forward_wr) ttx_psw_backward then yields d per_sample_weights for the whole batch;
 *  - ttx_cache_weighted_grad_n: scaled[n] = per_sample_weights[n] * grad_output[rowidx[n]], iota[n] = n for the
 *    cached entries: ttx_cache_backward_*_n(.., grad_output = scaled, rowidx = iota, ..) is the weighted backward. */
int ttx_preprocess_indices_async_w(int64_t nnz, const int64_t* colidx, int64_t num_bags_total, const int64_t* offsets,
                                   int32_t num_tables, int32_t warmup, int64_t hashtbl_size, const int64_t* hashtbl,
                                   const int32_t* cache_state, int64_t* rowidx, int64_t* tableidx,
                                   int64_t* partitioned_colidx, int64_t* partitioned_rowidx, int32_t* cache_locations,
                                   int32_t* num_tt_host, int32_t* partitioned_host, int32_t* num_tt_dev,
                                   int64_t* upd_hashtbl, int64_t* upd_cache_freq, const float* per_sample_weights,
                                   float* partitioned_weights, int32_t* partitioned_origin, void* workspace,
                                   size_t workspace_bytes, ttx_stream_t stream);
int ttx_cache_forward_nw(int32_t B, int64_t nnz, const int32_t* skip_dev, const int32_t* cache_locations,
                         const int64_t* rowidx, const float* per_sample_weights, int32_t D, const float* cache_weight,
                         float* output, ttx_stream_t stream);
int ttx_cache_rows_n(int64_t nnz, const int32_t* skip_dev, const int32_t* cache_locations, int32_t D,
                     const float* cache_weight, float* rows, ttx_stream_t stream);
int ttx_cache_weighted_grad_n(int64_t nnz, const int32_t* skip_dev, int32_t D, const float* grad_output,
                              const int64_t* rowidx, const float* per_sample_weights, float* scaled, int64_t* iota,
                              ttx_stream_t stream);

/* Lookup prologue of a batch while the cache is not live (warmup): what the module does
 * before the contraction -- update_cache_state (tt_embeddings_ops.py:827-833) when
 * upd_hashtbl / upd_cache_freq are non-NULL, preprocess_indices_sync with warmup = true
 * (:834-846: rowidx / tableidx from the offsets) and ttx_plan_build -- as ONE launch when the
 * batch qualifies (one table, every tables*p_t <= 256, 1024 < nnz <= 16384, <= 4096 bags),
 * otherwise as the separate launches; results are identical either way. */
int ttx_lookup_prologue(const ttx_geom* g, int64_t nnz, const int64_t* colidx,
                        int64_t num_bags_total, const int64_t* offsets, int64_t hashtbl_size,
                        int64_t* upd_hashtbl, int64_t* upd_cache_freq, int64_t* rowidx,
                        int64_t* tableidx, void* plan, size_t plan_bytes, ttx_stream_t stream);
/* The same with the number of LIVE lookups on the device (round 5; not in the reference): colidx holds nnz entries, the first
 * *nnz_dev of them (offsets[nb] == *nnz_dev <= nnz) are the batch -- only they are planned, and every call that later takes this
 * plan (ttx_tt_forward*, ttx_tt_backward*) works on exactly those, whatever nnz its launches are sized by.  nnz_dev == NULL: all
 * nnz.  For a table-sharded owner of RAGGED bags: fixed-capacity exchange buffers, no host read-back of the count.
 * (With a frequency table -- H > 0 -- only on the one-launch route; otherwise TTX_EUNSUPPORTED.) */
int ttx_lookup_prologue_n(const ttx_geom* g, int64_t nnz, const int64_t* colidx, int64_t nb, const int64_t* offsets,
                          int64_t H, int64_t* upd_hashtbl, int64_t* upd_cache_freq, int64_t* rowidx,
                          int64_t* tableidx, void* plan, size_t plan_bytes, const int32_t* nnz_dev, ttx_stream_t stream);

/* The prologues of SEVERAL batches in one launch (plan a round of training batches ahead: a batch's prologue depends on
 * its indices only, not on the cores).  colidx_host / offsets_host: HOST arrays of nbatch device pointers, every batch
 * with nnz indices and nb + 1 offsets; rowidx / tableidx: [nbatch][nnz]; plans: nbatch plan buffers plan_stride bytes
 * apart (a multiple of 256, >= ttx_plan_bytes(g, nnz)).  Batch z's results are exactly ttx_lookup_prologue's for that
 * batch; the frequency table counts all batches (counts commute).  One launch per 16 batches when the batch qualifies
 * for the single-launch prologue, the per-batch calls otherwise. */
int ttx_lookup_prologue_multi(const ttx_geom* g, int32_t nbatch, int64_t nnz, const int64_t* const* colidx_host,
                              int64_t num_bags_total, const int64_t* const* offsets_host, int64_t hashtbl_size,
                              int64_t* upd_hashtbl, int64_t* upd_cache_freq, int64_t* rowidx, int64_t* tableidx,
                              void* plans, size_t plan_stride, ttx_stream_t stream);

/* The same for a LIVE cache (one table; tt_embeddings_ops.py:827-846 with self.warmup == False, then the plan of the
 * misses): per batch z exactly ttx_preprocess_indices_async(num_tables = 1, warmup = 0, frequency update on the same
 * table, split point to num_tt_dev[z]) followed by ttx_plan_build_n(.., num_tt_dev + z, pcol_z, tableidx_z, prow_z) --
 * as three launches per 16 batches when the batch qualifies for the single-launch plan (every p_t <= 256, nnz <= 16384),
 * batch after batch otherwise.  rowidx / tableidx / pcol / prow / ploc: [nbatch][nnz]; num_tt_dev: [nbatch].  A cached
 * lookup's location depends on (hashtbl, cache_state) as the last ttx_cache_populate left them, not on the frequency
 * counts, so batch z's results do not depend on which other batches were counted first; they are void once the cache
 * is populated again. */
size_t ttx_lookup_prologue_cached_multi_workspace_bytes(int32_t nbatch, int64_t nnz);
int ttx_lookup_prologue_cached_multi(const ttx_geom* g, int32_t nbatch, int64_t nnz, const int64_t* const* colidx_host,
                                     int64_t num_bags, const int64_t* const* offsets_host, int64_t hashtbl_size,
                                     int64_t* hashtbl, int64_t* cache_freq, const int32_t* cache_state,
                                     int64_t* rowidx, int64_t* tableidx, int64_t* partitioned_colidx,
                                     int64_t* partitioned_rowidx, int32_t* cache_locations, int32_t* num_tt_dev,
                                     void* plans, size_t plan_stride, void* workspace, size_t workspace_bytes,
                                     ttx_stream_t stream);

/* replaces cache_populate_cuda (tt_embeddings.cpp:76-86,
 * tt_embeddings_cuda.cu:1260-1336): stable descending radix sort of the slots
 * by frequency, the top cache_size keys get cache rows (cache_state[slot] =
 * rank), the others are evicted from the table, and the cache rows are
 * decompressed from the TT cores. */
size_t ttx_cache_populate_workspace_bytes(const ttx_geom* g, int64_t hashtbl_size,
                                          int64_t cache_size, int32_t D);
int ttx_cache_populate(const ttx_geom* g, const float* const* tt_cores,
                       int64_t hashtbl_size, int64_t* hashtbl,
                       int64_t* cache_freq, int32_t* cache_state,
                       int64_t cache_size, int32_t D, float* cache_weight,
                       void* workspace, size_t workspace_bytes,
                       ttx_stream_t stream);
/* The same with per-call behaviour flags (round 6: the process-wide ttx_set_reference_exact of rounds 3-5 is gone).
 * TTX_POPULATE_REFERENCE_EXACT: cache_state[slot] of an EVICTED slot is left untouched, exactly as the reference's
 * mark_popular_colidx_kernel does (tt_embeddings_cuda.cu:1131-1133).  Default (flags = 0 = ttx_cache_populate): the slot's
 * cache row is dropped (cache_state = -1) -- otherwise, after a second populate, the next key inserted into that slot is
 * served, and trains, another index's cached row.  Identical on a first populate. */
#define TTX_POPULATE_REFERENCE_EXACT 1
int ttx_cache_populate_f(const ttx_geom* g, const float* const* tt_cores,
                         int64_t hashtbl_size, int64_t* hashtbl,
                         int64_t* cache_freq, int32_t* cache_state,
                         int64_t cache_size, int32_t D, float* cache_weight, int32_t flags,
                         void* workspace, size_t workspace_bytes,
                         ttx_stream_t stream);

/* replaces cache_forward_cuda (tt_embeddings.cpp:97-103,
 * tt_embeddings_cuda.cu:1498-1572): output[rowidx[n], :] += cache_weight[
 * cache_locations[n], :], summed per run of equal rowidx. */
int ttx_cache_forward(int32_t B, int64_t nnz, const int32_t* cache_locations,
                      const int64_t* rowidx, int32_t D,
                      const float* cache_weight, float* output,
                      ttx_stream_t stream);

/* replaces cache_backward_sgd_cuda (tt_embeddings.cpp:105-111,
 * tt_embeddings_cuda.cu:1574-1657):
 * cache_weight[loc[n], :] += -grad_output[rowidx[n], :] * lr. */
int ttx_cache_backward_sgd(int64_t nnz, int32_t D, const float* grad_output,
                           const int32_t* cache_locations, const int64_t* rowidx,
                           float learning_rate, float* cache_weight,
                           ttx_stream_t stream);

/* replaces cache_backward_dense_cuda (tt_embeddings.cpp:113-119,
 * tt_embeddings_cuda.cu:1659-1733): grad_cache_weight[cache_size, D] is
 * overwritten with the scatter-sum of grad_output rows. */
int ttx_cache_backward_dense(int64_t nnz, int32_t D, const float* grad_output,
                             const int32_t* cache_locations,
                             const int64_t* rowidx, int64_t cache_size,
                             float* grad_cache_weight, ttx_stream_t stream);

/* replaces cache_backward_rowwise_adagrad_approx_cuda (tt_embeddings.cpp:121-129,
 * tt_embeddings_cuda.cu:1735-1835): per bag g2 = mean(g*g); per cached lookup
 * old = state[loc] (then state[loc] += g2), w[loc,:] -= g * lr/(sqrt(old+g2)+eps).
 * Bags are processed in order (the reference leaves the order of two bags that
 * hit the same cache row to the hardware). */
int ttx_cache_backward_rowwise_adagrad_approx(
    int64_t nnz, int32_t D, const float* grad_output,
    const int32_t* cache_locations, const int64_t* rowidx, float learning_rate,
    float eps, float* cache_optimizer_state, float* cache_weight,
    ttx_stream_t stream);

/* ------------------------------------------------------------- profiling ---
 * Live kernel timing for bench.py's roofline block: ttx_profile_enable(mask)
 * selects kernel slots (bit w = slot w, 0 = off); every launch of a selected
 * kernel is bracketed by two HIP events on the launch stream.
 * ttx_profile_read synchronises the pending events and returns, for kernel
 * `which` (0 = forward contraction, 1 = backward contraction, 2 = reduce/apply,
 * 3 = plan, 4 = bag pooling, 5 = cache gather fwd), the launch count and the
 * summed duration in milliseconds since the last reset. */
#define TTX_PROF_FWD 0
#define TTX_PROF_BWD 1
#define TTX_PROF_APPLY 2
#define TTX_PROF_PLAN 3
#define TTX_PROF_POOL 4
#define TTX_PROF_CACHE_FWD 5
#define TTX_PROF_NUM 6
int ttx_profile_enable(int mask);
/* the same without reading back pending event pairs: for launches recorded inside a hipGraph capture, whose
 * events only carry times once the graph has been replayed (read them with ttx_profile_read afterwards) */
int ttx_profile_mask(int mask);
int ttx_profile_reset(void);
int ttx_profile_read(int which, int64_t* launches, double* total_ms);

/* The cache rows' update of one batch WITHOUT atomics: deterministic (bit-identical from run to run), one writer per cache row.
 * Replaces cache_backward_sgd_cuda / cache_backward_dense_cuda / cache_backward_rowwise_adagrad_approx_cuda
 * (tt_embeddings.cpp:105-129, tt_embeddings_cuda.cu:1574-1835) for callers that give it a workspace: the cached lookups
 * [*skip_dev, nnz) (skip_dev NULL: all) are grouped by cache row with a stable sort, a row's bag gradients are added in INDEX
 * order and applied once.  optim: TTX_OPTIM_SGD (dst = cache_weight, += -lr * sum), TTX_OPTIM_DENSE (dst = the cache_weight gradient
 * [cache_size, D], zeroed here) or TTX_OPTIM_ADAGRAD (dst = cache_weight, cache_optimizer_state [cache_size]: the reference's
 * per-lookup sequence old = state, state += g2, w -= g * lr / (sqrt(old + g2) + eps), taken in index order within a row -- the
 * order of a sequential execution, where the reference's depends on which warp arrives first).  num_bags: rows of grad
 * (row-wise Adagrad).  Up to 32,768 lookups (D % 4 == 0, D <= 256) it is ONE launch -- every work-group owns the rows
 * row % G == g, finds and groups them in LDS -- and needs no workspace; beyond that a chain of ~16 launches (stable radix sort, run
 * heads, ordered sums, apply).  The atomic entry points above stay: one launch, faster below ~300k cached lookups (row-wise
 * Adagrad: ~60k). */
size_t ttx_cache_backward_sorted_workspace_bytes(int64_t nnz, int64_t num_bags, int32_t D);
int ttx_cache_backward_sorted(int32_t optim, int64_t nnz, const int32_t* skip_dev, int64_t num_bags, int32_t D,
                              const float* grad_output, const int32_t* cache_locations, const int64_t* rowidx,
                              float learning_rate, float eps, int64_t cache_size, float* cache_optimizer_state,
                              float* dst, void* workspace, size_t workspace_bytes, ttx_stream_t stream);

/* test entry: the stable descending 64-bit radix sort of (key, value) pairs behind ttx_cache_populate, on its own
 * (what the reference asks of cub::DeviceRadixSort::SortPairsDescending, tt_embeddings_cuda.cu:1280-1308).  Stateless. */
size_t ttx_debug_sort_workspace_bytes(int64_t n);
int ttx_debug_sort_pairs_desc(int64_t n, const int64_t* keys, const int64_t* vals, int64_t* keys_out, int64_t* vals_out,
                              void* workspace, size_t workspace_bytes, ttx_stream_t stream);
/* host-side query, stateless: the walk the generic kernels take for geometry g:
 * out[6] = {lookups per chunk, q1 blocks per column pass, rows per K block, column passes, K blocks, LDS bytes};
 * all zero when a shape-specialised kernel takes the geometry (or nothing fits the LDS). */
int ttx_debug_tiles(const ttx_geom* g, int32_t* out);
/* Test / ablation knobs (skip kernel phases, force the generic kernels, LDS budget, chunk size, stamps): NOT in this library.
 * They exist in the test build of the same sources only (-DTTX_TEST_HOOKS -> libttx_hooks.so; include/ttx_test_hooks.h);
 * libttx.so compiles every knob as a constant at its default and exports no setter.  ttx_has_test_hooks() tells the builds
 * apart (0 = product); ttx_debug_state() is the bit mask of knobs away from their defaults -- always 0 in the product build;
 * bench.py refuses to time a library where it is not. */
int ttx_has_test_hooks(void);
int ttx_debug_state(void);
int ttx_cache_debug_state(void); /* (internal helper of the above) */

#ifdef __cplusplus
}
#endif
#endif /* TTX_H_ */
