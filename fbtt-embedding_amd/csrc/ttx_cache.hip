// ttx_cache.hip -- LFU software cache of decompressed rows: hash table
// insert / lookup, stable partition, cache-row gather forward / backward,
// populate (radix sort + mark/evict + decompress).  gfx950, wave64.
//
// Replaces tt_embeddings_cuda.cu:1077-1835 and hashtbl_cuda_utils.cuh, and the
// two CUB device algorithms the reference calls (DeviceRadixSort::
// SortPairsDescending cu:1281-1307, DevicePartition::Flagged cu:1437-1479) with
// wave-ballot ranking: a "wave unit" (one wavefront walking a contiguous range
// in order) ranks 64 keys at a time with wave_match8 / ballot + popcount, keeps
// its running digit offsets in LDS, and unit totals are combined by a single
// exclusive scan -- stable, atomic-free and deterministic.
#include "ttx_internal.h"

namespace ttx {

constexpr int kCT = 256;

// hashtbl_cuda_utils.cuh:135-154 (incl. the early-out on the SEARCH key, :146)
__device__ __forceinline__ int32_t hashtbl_find(int64_t key, int32_t size, const int64_t* keys) {
  int32_t idx = (int32_t)hash64(key, size);
#pragma unroll
  for (int c = 0; c < kMaxProbes; ++c) {
    if (key == keys[idx]) return idx;
    else if (key == -1) return -1;
    idx = (idx + 1) % size;
  }
  return -1;
}

// hashtbl_cuda_utils.cuh:102-133 with accumulate == true: 64-bit CAS claims the
// slot, 64-bit atomic add bumps the frequency.
__global__ __launch_bounds__(kCT) void update_cache_state_kernel(
    int64_t N, const int64_t* __restrict__ colidx, int32_t H, int64_t* hashtbl, int64_t* cache_freq) {
  const int64_t n = (int64_t)blockIdx.x * kCT + threadIdx.x;
  hashtbl_count_wave(n < N ? colidx[n] : 0, n < N, H, hashtbl, cache_freq);
}

// rowidx/tableidx of every bag AND, per index, the frequency update and / or the cache lookup, in one
// launch (tiny kernels cost ~4 us each on this chip whatever they do): threads [0, 8*nb) walk the
// bags 8 lanes per bag, threads [0, N) each handle one index.
//   upd_hashtbl != NULL : update_cache_state (cu:1077-1113) on (upd_hashtbl, cache_freq)
//   loc != NULL         : cache_lookup_kernel (cu:1356-1375): loc[i] = cache row or -1 (TT entry), and
//                         unit_cnt[blockIdx.x] = TT entries among the work-group's kCT positions
__global__ __launch_bounds__(kCT) void rowidx_update_kernel(int64_t nb, int32_t B,
                                                           const int64_t* __restrict__ offsets,
                                                           int64_t* rowidx, int64_t* tableidx, int64_t N,
                                                           const int64_t* __restrict__ colidx, int32_t H,
                                                           int64_t* upd_hashtbl, int64_t* cache_freq,
                                                           const int64_t* __restrict__ hashtbl,
                                                           const int32_t* __restrict__ cache_state, int32_t* loc,
                                                           int* unit_cnt, ProBatch mb, long long ws_stride) {
  __shared__ int wc[kCT / kWave];
  if (blockIdx.y > 0) {  // batch z of ttx_lookup_prologue_cached_multi (ws_stride: ints between the batches' loc / unit_cnt)
    const int z = blockIdx.y;
    TTX_PICK_BATCH(mb, z, colidx, offsets)
    rowidx += (long long)z * mb.out_stride;
    tableidx += (long long)z * mb.out_stride;
    if (loc) { loc += (long long)z * ws_stride; unit_cnt += (long long)z * ws_stride; }
  }
  const int64_t gt = (int64_t)blockIdx.x * kCT + threadIdx.x;
  const int64_t b = gt >> 3;
  if (b < nb) {
    const int64_t beg = offsets[b], end = offsets[b + 1];
    for (int64_t l = beg + (threadIdx.x & 7); l < end; l += 8) {
      rowidx[l] = b % B;
      tableidx[l] = b / B;
    }
  }
  const bool valid = gt < N;
  const int64_t key = valid ? colidx[gt] : 0;
  // The reference updates the table in one launch and looks the batch up in the next (cu:1077-1113, then :1356-1375): a look-up
  // sees EVERY insert of its batch.  That matters for a cached key that sits at its second or third probe since populate emptied
  // the slot before it: the batch's own count re-inserts the key into the empty slot, the look-up finds it THERE (cache_state -1)
  // and the key is a TT lookup from then on.  In one launch a plain find would race with the other copies' inserts (hit or miss
  // by timing -- found as run-to-run differences of a free-running loop, round 5); the slot the count landed in is what the
  // find of the reference's second launch returns, for every copy of the key alike.
  int32_t counted = -1;
  if (upd_hashtbl) counted = hashtbl_count_wave(key, valid, H, upd_hashtbl, cache_freq);
  if (loc) {
    bool tt = false;
    if (valid) {
      const int32_t slot = (upd_hashtbl && upd_hashtbl == hashtbl && key != -1) ? counted : hashtbl_find(key, H, hashtbl);
      int32_t cl = -1;
      if (slot != -1) cl = cache_state[slot];
      tt = (cl == -1);
      loc[gt] = cl;
    }
    const int n = __popcll(__ballot(tt));
    if (lane_id() == 0) wc[threadIdx.x / kWave] = n;
    __syncthreads();
    if (threadIdx.x == 0 && (int64_t)blockIdx.x * kCT < N) {
      int s = 0;
      for (int k = 0; k < kCT / kWave; ++k) s += wc[k];
      unit_cnt[blockIdx.x] = s;
    }
  }
}

__global__ void set_int_kernel(int32_t* p, int32_t v) { *p = v; }

// compute_rowidx_kernel, cu:1338-1354: one 8-lane group per bag
__global__ __launch_bounds__(kCT) void compute_rowidx_kernel(int64_t nb, int32_t B,
                                                            const int64_t* __restrict__ offsets,
                                                            int64_t* rowidx, int64_t* tableidx) {
  const int64_t b = (int64_t)blockIdx.x * (kCT / 8) + threadIdx.x / 8;
  if (b >= nb) return;
  const int64_t beg = offsets[b], end = offsets[b + 1];
  for (int64_t l = beg + (threadIdx.x & 7); l < end; l += 8) {
    rowidx[l] = b % B;
    tableidx[l] = b / B;
  }
}

// ---- core-0 row split (ttx_split0_expand, include/ttx.h): a table whose first factor q0 = k q0' is looked up as k
// "part lookups" per index in a table with p0' = k p0 rows of q0' -- core 0 [p0, q0, r1] IS [k p0, q0', r1] -- part h of bag b
// becoming bag k b + h of a batch with k times the bags and D / k columns: [B, D] row-major is [k B, D / k].
// One 8-lane group per bag (compute_rowidx_kernel's shape): virtual offsets, and the bag's indices copied k times.
__global__ __launch_bounds__(kCT) void split0_expand_kernel(int64_t nb, int32_t k, int64_t p_rest,
                                                           const int64_t* __restrict__ indices,
                                                           const int64_t* __restrict__ offsets, int64_t* out_idx,
                                                           int64_t* out_off) {
  const int64_t b = (int64_t)blockIdx.x * (kCT / 8) + threadIdx.x / 8;
  if (b >= nb) return;
  const int64_t beg = offsets[b], end = offsets[b + 1], len = end - beg;
  const int l8 = threadIdx.x & 7;
  if (l8 < k || k > 8)
    for (int h = l8; h < k; h += 8) out_off[b * k + h] = beg * k + h * len;
  if (b == nb - 1 && l8 == 0) out_off[nb * k] = end * k;
  for (int64_t l = beg + l8; l < end; l += 8) {
    const int64_t idx = indices[l];
    const int64_t i0 = idx / p_rest;                  // (i0 p_rest + rem  ->  (k i0 + h) p_rest + rem)
    const int64_t v0 = idx + i0 * (k - 1) * p_rest;
    for (int h = 0; h < k; ++h) out_idx[beg * k + h * len + (l - beg)] = v0 + h * p_rest;
  }
}

// ---- stable partition (cache_lookup_kernel cu:1356-1375 + Flagged) ---------
// rowidx_update_kernel looked every index up (loc, -1 <=> TT entry) and counted the TT entries of
// every unit = work-group of kCT positions.  Few units: the scatter launch sums the counts before
// its own itself; many: scan_units_kernel turns them into exclusive prefixes first.
__global__ __launch_bounds__(1024) void scan_units_kernel(int U, int* unit_cnt, int* total_out) {
  __shared__ int wt[17];
  int carry = 0;
  for (int b0 = 0; b0 < U; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int v = i < U ? unit_cnt[i] : 0;
    const int inc = wave_incl_scan(v);
    const int w = threadIdx.x / kWave;
    if (lane_id() == kWave - 1) wt[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
      int run = 0;
      for (int k = 0; k < 16; ++k) { int c = wt[k]; wt[k] = run; run += c; }
      wt[16] = run;
    }
    __syncthreads();
    if (i < U) unit_cnt[i] = carry + wt[w] + inc - v;
    carry += wt[16];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(kCT) void partition_scatter_kernel(
    int N, int U, int scanned, const int64_t* __restrict__ colidx, const int64_t* __restrict__ rowidx,
    const int32_t* __restrict__ loc, const int* __restrict__ unit_cnt, int64_t* pcol, int64_t* prow,
    int32_t* ploc, int* total_out, int32_t* num_tt_dev, ProBatch mb, long long ws_stride,
    const float* __restrict__ psw, float* ppsw, int32_t* porig) {
  __shared__ int wsum[kCT / kWave], wbase[kCT / kWave];
  if (blockIdx.y > 0) {  // batch z of ttx_lookup_prologue_cached_multi
    const int z = blockIdx.y;
    const int64_t* unused = nullptr;
    TTX_PICK_BATCH(mb, z, colidx, unused)
    (void)unused;
    const long long sh = (long long)z * mb.out_stride;
    rowidx += sh; pcol += sh; prow += sh; ploc += sh;
    loc += (long long)z * ws_stride; unit_cnt += (long long)z * ws_stride; total_out += (long long)z * ws_stride;
    if (num_tt_dev) num_tt_dev += z;
  }
  const int g = blockIdx.x, tid = threadIdx.x, lane = lane_id(), w = tid / kWave;
  int part = 0;  // TT entries before this unit
  if (scanned) {
    part = (tid == 0) ? unit_cnt[g] : 0;
  } else {
    for (int u = tid; u < g; u += kCT) part += unit_cnt[u];
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) part += __shfl_xor(part, o, kWave);
  const int i = g * kCT + tid;
  const bool valid = i < N;
  const int32_t cl = valid ? loc[i] : 0;
  const bool tt = valid && cl == -1;
  const unsigned long long m = __ballot(tt);
  if (lane == 0) { wsum[w] = __popcll(m); wbase[w] = part; }
  __syncthreads();
  int run = 0, own = 0;
  for (int k = 0; k < kCT / kWave; ++k) { run += wbase[k]; if (k < w) run += wsum[k]; own += wsum[k]; }
  if (g == U - 1 && tid == 0) {  // the split point, for the host copy and for the device-side consumers
    int base = 0;
    for (int k = 0; k < kCT / kWave; ++k) base += wbase[k];
    *total_out = base + own;
    if (num_tt_dev) *num_tt_dev = base + own;
  }
  if (valid) {
    const int tt_before = run + __popcll(m & lanemask_lt());
    // selected keep their order at the front; rejected go to the rear, reversed
    const int dst = tt ? tt_before : (N - 1 - (i - tt_before));
    pcol[dst] = colidx[i];
    prow[dst] = rowidx[i];
    ploc[dst] = cl;
    if (psw) ppsw[dst] = psw[i];  // nn.EmbeddingBag per_sample_weights travel with their lookups ...
    if (porig) porig[dst] = i;    // ... and the way back (the weights' gradient is returned in the caller's order)
  }
}

// ---- cache row gather / scatter --------------------------------------------
// cache_forward_kernel cu:1498-1538: run heads add their run's cache rows, in
// index order, onto the current output value.  32 lanes per lookup.  HBM/latency bound
// (268 B per cached lookup): the rows of a run are fetched 8 at a time -- 8 cache locations,
// then 8 x EPL independent row loads in flight per lane -- and added in index order.
template <int EPL>  // elements per lane = ceil(D / 32), 1..4
__device__ __forceinline__ void gather_run(int n, int sl, int D, int l, const int32_t* __restrict__ loc,
                                           const float* __restrict__ w, float* o) {
  float acc[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) acc[k] = (l + 32 * k < D) ? o[l + 32 * k] : 0.f;
  for (int j0 = 0; j0 < sl; j0 += 8) {
    int lc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) lc[u] = (j0 + u < sl) ? loc[n + j0 + u] : -1;
    float v[8][EPL];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < EPL; ++k)
        v[u][k] = (lc[u] >= 0 && l + 32 * k < D) ? w[(size_t)lc[u] * D + l + 32 * k] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (j0 + u < sl) {
#pragma unroll
        for (int k = 0; k < EPL; ++k) acc[k] += v[u][k];
      }
  }
#pragma unroll
  for (int k = 0; k < EPL; ++k)
    if (l + 32 * k < D) o[l + 32 * k] = acc[k];
}

// (skip_dev, all three cache kernels: device-side count of leading entries that are NOT cached --
// the split point of the partition -- so that no host read-back is needed; NULL = 0)
__global__ __launch_bounds__(kCT) void cache_forward_kernel(int N, int D, const int* __restrict__ skip_dev,
                                                           const int64_t* __restrict__ rowidx,
                                                           const int32_t* __restrict__ loc,
                                                           const float* __restrict__ w, float* out) {
  const int n = blockIdx.x * (kCT / 32) + threadIdx.x / 32;
  const int l = threadIdx.x & 31;
  if (skip_dev) { const int k = max(0, min(N, *skip_dev)); N -= k; rowidx += k; loc += k; }
  if (n >= N) return;
  const int64_t r = rowidx[n];
  if (n > 0 && rowidx[n - 1] == r) return;
  // run length: the 32 lanes test 32 candidates per step (one ballot) instead of walking them
  const unsigned long long half = (threadIdx.x & 32) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
  int sl = 1;
  for (;;) {
    const int c = n + sl + l;
    const bool same = c < N && rowidx[c] == r;
    const unsigned long long m = (__ballot(!same) & half) >> (threadIdx.x & 32);
    if (m) { sl += __builtin_ctzll(m); break; }
    sl += 32;
  }
  float* o = out + (size_t)r * D;
  if (D <= 32) return gather_run<1>(n, sl, D, l, loc, w, o);
  if (D <= 64) return gather_run<2>(n, sl, D, l, loc, w, o);
  if (D <= 128) return gather_run<4>(n, sl, D, l, loc, w, o);
  for (int e = l; e < D; e += 32) {
    float acc = o[e];
    for (int j = 0; j < sl; ++j) acc += w[(size_t)loc[n + j] * D + e];
    o[e] = acc;
  }
}

// ... with nn.EmbeddingBag's per_sample_weights: out[row] += sum_j psw[n + j] * w[loc[n + j]], in index order (one
// 32-lane group per run head; the weighted form is the rare one and stays simple)
__global__ __launch_bounds__(kCT) void cache_forward_w_kernel(int N, int D, const int* __restrict__ skip_dev,
                                                             const int64_t* __restrict__ rowidx,
                                                             const int32_t* __restrict__ loc,
                                                             const float* __restrict__ psw,
                                                             const float* __restrict__ w, float* out) {
  const int n = blockIdx.x * (kCT / 32) + threadIdx.x / 32;
  const int l = threadIdx.x & 31;
  if (skip_dev) { const int k = max(0, min(N, *skip_dev)); N -= k; rowidx += k; loc += k; psw += k; }
  if (n >= N) return;
  const int64_t r = rowidx[n];
  if (n > 0 && rowidx[n - 1] == r) return;
  const unsigned long long half = (threadIdx.x & 32) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
  int sl = 1;
  for (;;) {
    const int c = n + sl + l;
    const bool same = c < N && rowidx[c] == r;
    const unsigned long long m = (__ballot(!same) & half) >> (threadIdx.x & 32);
    if (m) { sl += __builtin_ctzll(m); break; }
    sl += 32;
  }
  float* o = out + (size_t)r * D;
  for (int e = l; e < D; e += 32) {
    float acc = o[e];
    for (int j = 0; j < sl; ++j) acc = fmaf(psw[n + j], w[(size_t)loc[n + j] * D + e], acc);
    o[e] = acc;
  }
}

// rows[n] = w[loc[n]] for the cached entries (behind the split point): the forward's rows of the hits, kept for the
// gradient of the per_sample_weights (ttx_psw_backward reads them like the contraction's rows of the misses)
__global__ __launch_bounds__(kCT) void cache_rows_gather_kernel(int N, int D, const int* __restrict__ skip_dev,
                                                               const int32_t* __restrict__ loc,
                                                               const float* __restrict__ w, float* __restrict__ rows) {
  const int k = skip_dev ? max(0, min(N, *skip_dev)) : 0;
  const int n = k + blockIdx.x * (kCT / 32) + threadIdx.x / 32;
  const int l = threadIdx.x & 31;
  if (n >= N) return;
  const float* src = w + (size_t)loc[n] * D;
  float* dst = rows + (size_t)n * D;
  for (int e = l; e < D; e += 32) dst[e] = src[e];
}

// scaled[n] = psw[n] * grad[rowidx[n]] and iota[n] = n for the cached entries: the cache backward kernels then take
// (scaled, iota) in place of (grad_output, rowidx) -- every lookup brings its own weighted gradient row
__global__ __launch_bounds__(kCT) void cache_scale_grad_kernel(int N, int D, const int* __restrict__ skip_dev,
                                                              const float* __restrict__ grad,
                                                              const int64_t* __restrict__ rowidx,
                                                              const float* __restrict__ psw,
                                                              float* __restrict__ scaled, int64_t* __restrict__ iota) {
  const int k = skip_dev ? max(0, min(N, *skip_dev)) : 0;
  const int n = k + blockIdx.x * (kCT / 32) + threadIdx.x / 32;
  const int l = threadIdx.x & 31;
  if (n >= N) return;
  const float* g = grad + (size_t)rowidx[n] * D;
  const float wn = psw[n];
  float* dst = scaled + (size_t)n * D;
  for (int e = l; e < D; e += 32) dst[e] = wn * g[e];
  if (l == 0) iota[n] = n;
}

// The same sums for D % 4 == 0, wave64-native (the 32-lane-group kernel above keeps 1 group in 20 busy at 20 lookups
// per bag: the others are not run heads and leave).  A wave owns 64 consecutive cached lookups: one coalesced load of
// their bag rows and cache locations, run heads by one ballot, and the wave's four 16-lane groups take the runs
// round-robin -- a row of 64 floats is ONE 16-byte load per lane, a run is summed in index order by one group with
// eight rows in flight.  The last run of a span may go on behind it (its group keeps walking); a run that began in
// the previous span belongs to that span's wave.  Locations travel through LDS (a cross-lane read would need the
// source lane active: the groups' loops diverge).
__global__ __launch_bounds__(kCT) void cache_forward4_kernel(int N, int D4, const int* __restrict__ skip_dev,
                                                            const int64_t* __restrict__ rowidx,
                                                            const int32_t* __restrict__ loc,
                                                            const float4* __restrict__ w, float4* out) {
  __shared__ unsigned char hpos[kCT / kWave][kWave];
  __shared__ int lloc[kCT / kWave][kWave];
  __shared__ int lrow[kCT / kWave][kWave];
  const int wv = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  if (skip_dev) { const int k = max(0, min(N, *skip_dev)); N -= k; rowidx += k; loc += k; }
  const int n0 = (blockIdx.x * (kCT / kWave) + wv) * kWave;
  if (n0 >= N) return;  // (wave-uniform)
  const int n = n0 + lane;
  const int r = n < N ? (int)rowidx[n] : -1;
  lloc[wv][lane] = n < N ? loc[n] : 0;
  int rprev = __shfl_up(r, 1, kWave);
  if (lane == 0) rprev = n0 > 0 ? (int)rowidx[n0 - 1] : -1;
  const bool head = n < N && (n == 0 || rprev != r);
  const unsigned long long heads = __ballot(head);
  if (head) {
    const int k = __popcll(heads & ((1ull << lane) - 1ull));
    hpos[wv][k] = (unsigned char)lane;
    lrow[wv][k] = r;
  }
  const int nheads = __popcll(heads);
  const int g = lane >> 4, l = lane & 15, sh = lane & 48;
  for (int k = g; k < nheads; k += 4) {
    const int pos = hpos[wv][k];
    const int hn = n0 + pos;
    const int row = lrow[wv][k];
    int sl;
    if (k + 1 < nheads) {
      sl = hpos[wv][k + 1] - pos;
    } else {  // the span's last run may go on behind it: 16 candidates per ballot (one group gets here)
      sl = min(N, n0 + kWave) - hn;
      if (hn + sl < N)
        for (;;) {
          const int c = hn + sl + l;
          const bool same = c < N && (int)rowidx[c] == row;
          const unsigned m = (unsigned)(__ballot(!same) >> sh) & 0xffffu;
          if (m) { sl += __builtin_ctz(m); break; }
          sl += 16;
        }
    }
    const int inspan = kWave - pos;  // lookups of the run whose location sits in LDS
    float4* o = out + (size_t)row * D4;
    for (int e = l; e < D4; e += 16) {
      float4 acc = o[e];
      for (int j0 = 0; j0 < sl; j0 += 8) {
        int lc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u;
          lc[u] = j < sl ? (j < inspan ? lloc[wv][pos + j] : loc[hn + j]) : -1;
        }
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = lc[u] >= 0 ? w[(size_t)lc[u] * D4 + e] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + u < sl) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      o[e] = acc;
    }
  }
}

#include "ttx_cache_scatter.inc"
__global__ __launch_bounds__(kCT) void cache_scatter_add_kernel(int N, int D, float scale,
                                                               const int* __restrict__ skip_dev,
                                                               const float* __restrict__ grad,
                                                               const int32_t* __restrict__ loc,
                                                               const int64_t* __restrict__ rowidx,
                                                               float* dst, int nmain, int K) {
  cache_scatter_add_body((int)blockIdx.x, N, D, scale, skip_dev, grad, loc, rowidx, dst, nmain, K);
}

// cache_backward_rowwise_adagrad_approx_kernel cu:1735-1795.  One wave per
// lookup: g2 = mean(g^2) of its bag, old = atomicAdd(state[loc], g2),
// mult = lr / (sqrt(old + g2) + eps) with the reference's double intermediate,
// w[loc,:] -= g * mult (atomic: the sum over lookups is order independent; only
// the `old` each lookup observes depends on arrival order, as in the reference).
// Hot rows [0, K) (see cache_scatter_add_kernel: a skewed stream puts a sixth of the batch on row 0, and 64 atomics per
// lookup on one row serialise in the L2 -- 91 us at cfg3) are taken out of the per-lookup path: work-group (k, s) lists
// the lookups of row k in segment s IN INDEX ORDER, takes the segment's share of the row's state with ONE atomic
// (old_s = atomicAdd(state[k], sum of the segment's g2)) and serves its lookups in index order from there
// (old_j = old_s + g2 of the segment's earlier lookups: one of the orders the per-lookup atomics could have arrived in,
// and the sequential order when the batch is one segment), then adds sum_j g_j * mult_j to the row once.
__device__ __forceinline__ float wave_incl_scan_f(float v) {
  const int lane = lane_id();
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const float u = __shfl_up(v, o, kWave);
    if (lane >= o) v += u;
  }
  return v;
}

__global__ __launch_bounds__(kCT) void cache_rowwise_adagrad_kernel(
    int N, int D, const int* __restrict__ skip_dev, const float* __restrict__ grad, const int32_t* __restrict__ loc,
    const int64_t* __restrict__ rowidx, float lr, float eps, float* state, float* wgt, int nmain, int K) {
  if (skip_dev) { const int k = max(0, min(N, *skip_dev)); N -= k; rowidx += k; loc += k; }
  if ((int)blockIdx.x >= nmain) {
    // ---- hot work-group (k, s) ----
    constexpr int kIt = kHotSeg / kCT, kW = kCT / kWave;
    __shared__ int list[kHotSeg];
    __shared__ float g2s[kHotSeg], mul[kHotSeg];
    __shared__ int cnt[kIt][kW];
    __shared__ float4 red[kCT];
    __shared__ float s_old;
    const int hb = blockIdx.x - nmain, k = hb % K, sbeg = (hb / K) * kHotSeg, tid = threadIdx.x;
    const int lane = lane_id(), w = tid / kWave;
    if (sbeg >= N) return;
    unsigned long long m[kIt];
    int myrow[kIt];
#pragma unroll
    for (int j = 0; j < kIt; ++j) {
      const int i = sbeg + j * kCT + tid;
      const bool hit = i < N && loc[i] == k;
      myrow[j] = hit ? (int)rowidx[i] : -1;
      m[j] = __ballot(hit);
      if (lane == 0) cnt[j][w] = __popcll(m[j]);
    }
    __syncthreads();
    int n = 0;
#pragma unroll
    for (int j = 0; j < kIt; ++j) {
#pragma unroll
      for (int ww = 0; ww < kW; ++ww) {
        if (j == 0 && ww == 0) n = 0;
        if (myrow[j] >= 0 && ww == w) list[n + __popcll(m[j] & lanemask_lt())] = myrow[j];  // (index order)
        n += cnt[j][ww];
      }
    }
    __syncthreads();
    if (n == 0) return;
    const int D4 = D / 4;
    const float4* g4 = (const float4*)grad;
    for (int e = tid; e < n; e += kCT) {  // g2 of every listed lookup's bag
      const float4* g = g4 + (size_t)list[e] * D4;
      float sq = 0.f;
      for (int x = 0; x < D4; ++x) {
        const float4 a = g[x];
        sq = fmaf(a.x, a.x, sq); sq = fmaf(a.y, a.y, sq); sq = fmaf(a.z, a.z, sq); sq = fmaf(a.w, a.w, sq);
      }
      g2s[e] = sq / D;
    }
    __syncthreads();
    if (w == 0) {  // exclusive prefix of g2 in list order, and the segment's one atomic on the row's state
      float run = 0.f;
      for (int b0 = 0; b0 < n; b0 += kWave) {
        const float v = b0 + lane < n ? g2s[b0 + lane] : 0.f;
        const float inc = wave_incl_scan_f(v);
        if (b0 + lane < n) mul[b0 + lane] = run + (inc - v);
        run += __shfl(inc, kWave - 1, kWave);
      }
      if (lane == 0) s_old = unsafeAtomicAdd(&state[k], run);
    }
    __syncthreads();
    const float old = s_old;
    for (int e = tid; e < n; e += kCT) {
      const float seen = old + mul[e];
      mul[e] = (float)(lr * (1.0 / (sqrtf(seen + g2s[e]) + eps)));
    }
    __syncthreads();
    const int parts = kCT / D4;  // thread = (float4 column, part); part p takes list entries p, p + parts, ..
    const int e4 = tid % D4, part = tid / D4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (part < parts) {
      for (int j = part; j < n; j += parts) {
        const float4 a = g4[(size_t)list[j] * D4 + e4];
        const float mj = mul[j];
        acc.x = fmaf(a.x, mj, acc.x); acc.y = fmaf(a.y, mj, acc.y); acc.z = fmaf(a.z, mj, acc.z); acc.w = fmaf(a.w, mj, acc.w);
      }
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < D4) {
      for (int p = 1; p < parts; ++p) {
        const float4 x = red[p * D4 + tid];
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
      float* wr = wgt + (size_t)k * D + (size_t)tid * 4;
      unsafeAtomicAdd(&wr[0], -acc.x);
      unsafeAtomicAdd(&wr[1], -acc.y);
      unsafeAtomicAdd(&wr[2], -acc.z);
      unsafeAtomicAdd(&wr[3], -acc.w);
    }
    return;
  }
  const int n = blockIdx.x * (kCT / kWave) + threadIdx.x / kWave;
  const int l = lane_id();
  if (n >= N) return;
  const int32_t c = loc[n];
  if (c < K) return;  // a hot row: the hot work-groups own it
  const float* g = grad + (size_t)rowidx[n] * D;
  float s = 0.f;
  for (int e = l; e < D; e += kWave) s = fmaf(g[e], g[e], s);
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
  const float g2 = s / D;
  float mult = 0.f;
  if (l == 0) {
    const float old = unsafeAtomicAdd(&state[c], g2);
    mult = (float)(lr * (1.0 / (sqrtf(old + g2) + eps)));
  }
  mult = __shfl(mult, 0, kWave);
  float* w = wgt + (size_t)c * D;
  for (int e = l; e < D; e += kWave) unsafeAtomicAdd(&w[e], -g[e] * mult);
}

// ---- multi-block stable LSD radix sort of (int64 key, int64 value) pairs ----
// descending on the key (digit' = 255 - digit), 8 bits per pass.  A wave unit
// walks WT consecutive elements; cnt is [256][U] (digit major).
__global__ __launch_bounds__(kCT) void radix_count_kernel(int N, int WT, int U, int shift,
                                                         const int64_t* __restrict__ keys, int* cnt) {
  __shared__ int hist[kCT / kWave][256];
  const int w = threadIdx.x / kWave, lane = lane_id();
  const int u = blockIdx.x * (kCT / kWave) + w;
  for (int e = lane; e < 256; e += kWave) hist[w][e] = 0;
  const int beg = u * WT;
  const int end = min(N, beg + WT);
  for (int base = beg; base < end; base += kWave) {
    const int i = base + lane;
    const bool valid = i < end;
    const unsigned dg = valid ? 255u - (unsigned)(((uint64_t)keys[i] >> shift) & 255u) : 0u;
    const unsigned long long peers = wave_match8(dg, valid);
    if (valid && (peers & lanemask_lt()) == 0) hist[w][dg] += __popcll(peers);
  }
  if (u < U)
    for (int e = lane; e < 256; e += kWave) cnt[(size_t)e * U + u] = hist[w][e];
}

// cnt is [256][U]: work-group e turns row e into its exclusive prefix (U <= 2048: two counts per thread) and leaves the
// row's total in tot[e]; the scatter launch adds the digits' bases itself.  (One work-group scanning all 256 * U counts
// one thread-strip after the other took 750 us per pass at 2^20 slots -- 2.3 of the populate's 2.9 ms.)
__global__ __launch_bounds__(1024) void radix_scan_kernel(int U, int* cnt, int* tot) {
  __shared__ int wt[17];
  int* row = cnt + (size_t)blockIdx.x * U;
  const int i0 = 2 * threadIdx.x, i1 = i0 + 1;
  const int a = i0 < U ? row[i0] : 0, b = i1 < U ? row[i1] : 0;
  const int s = a + b;
  const int inc = wave_incl_scan(s);
  const int w = threadIdx.x / kWave;
  if (lane_id() == kWave - 1) wt[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k < 16; ++k) { int c = wt[k]; wt[k] = run; run += c; }
    wt[16] = run;
  }
  __syncthreads();
  const int ex = wt[w] + inc - s;
  if (i0 < U) row[i0] = ex;
  if (i1 < U) row[i1] = ex + a;
  if (threadIdx.x == 0) tot[blockIdx.x] = wt[16];
}

__global__ __launch_bounds__(kCT) void radix_scatter_kernel(int N, int WT, int U, int shift,
                                                           const int64_t* __restrict__ keys,
                                                           const int64_t* __restrict__ vals,
                                                           const int* __restrict__ cnt,
                                                           const int* __restrict__ tot,
                                                           int64_t* okeys, int64_t* ovals) {
  __shared__ int run[kCT / kWave][256];
  __shared__ int dbase[256], wsum[kCT / kWave];
  const int w = threadIdx.x / kWave, lane = lane_id();
  const int u = blockIdx.x * (kCT / kWave) + w;
  {  // first position of every digit: exclusive prefix of the digit totals (kCT == 256 == digits)
    const int v = tot[threadIdx.x];
    const int inc = wave_incl_scan(v);
    if (lane == kWave - 1) wsum[w] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; ++k) base += wsum[k];
    dbase[threadIdx.x] = base + inc - v;
    __syncthreads();
  }
  if (u < U)
    for (int e = lane; e < 256; e += kWave) run[w][e] = dbase[e] + cnt[(size_t)e * U + u];
  const int beg = u * WT;
  const int end = min(N, beg + WT);
  for (int base = beg; base < end; base += kWave) {
    const int i = base + lane;
    const bool valid = i < end;
    int64_t k = 0;
    unsigned dg = 0;
    if (valid) {
      k = keys[i];
      dg = 255u - (unsigned)(((uint64_t)k >> shift) & 255u);
    }
    const unsigned long long peers = wave_match8(dg, valid);
    if (valid) {
      const int before = run[w][dg];
      const int pos = before + __popcll(peers & lanemask_lt());
      okeys[pos] = k;
      ovals[pos] = vals[i];
      if ((peers & lanemask_lt()) == 0) run[w][dg] = before + __popcll(peers);
    }
  }
}

__global__ __launch_bounds__(1024) void max_key_kernel(int N, const int64_t* __restrict__ keys,
                                                      unsigned long long* out) {  // *out starts at 0
  __shared__ unsigned long long wm[16];
  unsigned long long m = 0;
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < N; i += gridDim.x * 1024) {
    const unsigned long long k = (unsigned long long)keys[i];
    m = k > m ? k : m;
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    const unsigned long long v = __shfl_xor(m, o, kWave);
    m = v > m ? v : m;
  }
  if (lane_id() == 0) wm[threadIdx.x / kWave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) m = wm[k] > m ? wm[k] : m;
    if (m) atomicMax(out, m);
  }
}

// mark_popular_colidx_kernel cu:1115-1139
__global__ __launch_bounds__(kCT) void mark_popular_kernel(int32_t H, int64_t cache_size,
                                                          int64_t* sorted_keys, int64_t* hashtbl,
                                                          int64_t* cache_freq, int32_t* cache_state, int keep_state) {
  const int64_t n = (int64_t)blockIdx.x * kCT + threadIdx.x;
  if (n >= H) return;
  const int64_t key = sorted_keys[n];
  if (key != -1) {
    const int32_t slot = hashtbl_find(key, H, hashtbl);
    if (slot < 0) return;  // only after a repeated populate; the reference writes OOB
    if (n < cache_size) {
      cache_state[slot] = (int32_t)n;
    } else {
      hashtbl[slot] = -1;
      cache_freq[slot] = 0;
      // Deliberate fix (the reference leaves cache_state[slot] as it was): after a SECOND populate an evicted slot
      // would keep its old cache row number, and the next key inserted into that slot would be served -- and would
      // update -- another index's cached row.  Identical to the reference on a first populate (state is all -1).
      // keep_state: ttx_cache_populate_f(flags = TTX_POPULATE_REFERENCE_EXACT) -- the reference's behaviour, bit for bit, on demand.
      if (!keep_state) cache_state[slot] = -1;
    }
  } else if (n < cache_size) {
    sorted_keys[n] = 0;  // "a hack to use batch gemm"
  }
}

static void launch_scatter_add(int64_t nnz, int32_t D, float scale, const int32_t* skip_dev, const float* grad,
                               const int32_t* loc, const int64_t* rowidx, float* dst, hipStream_t st) {
  const int nmain = (int)((nnz + kCT / 32 - 1) / (kCT / 32));
  const int K = (((uintptr_t)grad & 15) == 0) ? hot_rows(nnz, D) : 0;
  const int nhot = K * (int)((nnz + kHotSeg - 1) / kHotSeg);
  hipLaunchKernelGGL(cache_scatter_add_kernel, dim3((unsigned)(nmain + nhot)), dim3(kCT), 0, st, (int)nnz, D, scale,
                     skip_dev, grad, loc, rowidx, dst, nmain, K);
}

static void unit_shape(long long n, int* WT, int* U) {  // wave units of the populate-time sorts
  long long wt = (n + 2047) / 2048;
  wt = (wt + kWave - 1) / kWave * kWave;
  if (wt < 256) wt = 256;
  *WT = (int)wt;
  *U = (int)((n + wt - 1) / wt);
  if (*U < 1) *U = 1;
}

constexpr int kScanUnits = 1024;  // beyond this many units the counts are scanned by their own launch
static int num_units(long long n) { return n > 0 ? (int)((n + kCT - 1) / kCT) : 1; }

// The 64-bit stable DESCENDING radix sort of (frequency, key) pairs behind cache_populate -- what the reference asks of
// cub::DeviceRadixSort::SortPairsDescending(cache_freq, hashtbl, bits [0, 64)) (tt_embeddings_cuda.cu:1280-1308): LSD, 8 bits
// per pass, as many passes as the largest frequency has bytes (one 8-byte read-back sizes it; a pass over all-zero digits
// would leave the order unchanged).  ws: 4 x align_up(8 H) + counts + 2 KB.  -> pointers to the sorted copies inside ws.
// passes: 8-bit passes to run (the caller knows how many bytes its keys have), or -1: one 8-byte read-back of the largest key
// sizes the sort (cache_populate; not capturable).
size_t sort_pairs_ws_bytes(int64_t n) {
  if (n <= 0) return 0;
  int WT, U;
  unit_shape(n, &WT, &U);
  return 4 * align_up((size_t)n * 8) + align_up((size_t)256 * U * 4) + 2048 + 256;
}

int sort_pairs_desc(int64_t H, const int64_t* keys_in, const int64_t* vals_in, char* ws, int64_t** keys_sorted,
                    int64_t** vals_sorted, hipStream_t st, int passes_known) {
  int WT, U;
  unit_shape(H, &WT, &U);
  const size_t hb = align_up((size_t)H * 8);
  int64_t* kA = (int64_t*)ws;
  int64_t* kB = (int64_t*)(ws + hb);
  int64_t* vA = (int64_t*)(ws + 2 * hb);
  int64_t* vB = (int64_t*)(ws + 3 * hb);
  int* cnt = (int*)(ws + 4 * hb);
  unsigned long long* dmax = (unsigned long long*)((char*)cnt + align_up((size_t)256 * U * 4));
  int* tot = (int*)(dmax + 32);  // 256 digit totals (the 2 KB behind the counts: dmax, then tot)
  const int N = (int)H;
  int passes = passes_known;
  if (passes < 0) {
    // size the sort: highest set bit of the largest frequency (8-byte read-back)
    TTX_HIP(hipMemsetAsync(dmax, 0, 8, st));
    hipLaunchKernelGGL(max_key_kernel, dim3((unsigned)((H + 1023) / 1024 < 256 ? (H + 1023) / 1024 : 256)), dim3(1024), 0, st, N,
                       keys_in, dmax);
    unsigned long long hmax = 0;
    TTX_HIP(hipMemcpyAsync(&hmax, dmax, 8, hipMemcpyDeviceToHost, st));
    TTX_HIP(hipStreamSynchronize(st));
    int bits = 0;
    while (bits < 64 && (hmax >> bits)) ++bits;
    passes = (bits + 7) / 8;
  }
  if (passes < 1) passes = 1;  // (>= 1: also produces the sorted copy)
  if (passes > 8) passes = 8;
  const unsigned blocks = (unsigned)((U + kCT / kWave - 1) / (kCT / kWave));
  const int64_t* ik = keys_in;
  const int64_t* iv = vals_in;
  int64_t* ok = kA;
  int64_t* ov = vA;
  for (int ps = 0; ps < passes; ++ps) {
    hipLaunchKernelGGL(radix_count_kernel, dim3(blocks), dim3(kCT), 0, st, N, WT, U, ps * 8, ik, cnt);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(1024), 0, st, U, cnt, tot);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(blocks), dim3(kCT), 0, st, N, WT, U, ps * 8, ik, iv, cnt, tot,
                       ok, ov);
    ik = ok;
    iv = ov;
    ok = (ok == kA) ? kB : kA;
    ov = (ov == vA) ? vB : vA;
  }
  TTX_HIP(hipGetLastError());
  if (keys_sorted) *keys_sorted = (int64_t*)ik;
  if (vals_sorted) *vals_sorted = (int64_t*)iv;
  return TTX_OK;
}


}  // namespace ttx

using namespace ttx;

extern "C" {

int ttx_update_cache_state(int64_t nnz, const int64_t* indices, int64_t H, int64_t* hashtbl,
                           int64_t* cache_freq, ttx_stream_t stream) {
  if (nnz == 0) return TTX_OK;  // cu:1095-1097
  if (H <= 0 || H >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "hashtbl_size=%lld must be in (0, 2^31)", (long long)H);
  if (!indices || !hashtbl || !cache_freq) TTX_FAIL(TTX_EINVAL, "NULL input");
  hipLaunchKernelGGL(update_cache_state_kernel, dim3((unsigned)((nnz + kCT - 1) / kCT)), dim3(kCT), 0,
                     (hipStream_t)stream, nnz, indices, (int32_t)H, hashtbl, cache_freq);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

int ttx_split0_expand(int64_t nnz, int64_t nb, int32_t k, int64_t p_rest, const int64_t* indices, const int64_t* offsets,
                      int64_t* out_indices, int64_t* out_offsets, ttx_stream_t stream) {
  if (nnz < 0 || nb <= 0 || k < 2 || k > 64 || p_rest <= 0) TTX_FAIL(TTX_EINVAL, "split0: nnz=%lld nb=%lld k=%d", (long long)nnz, (long long)nb, k);
  if (!offsets || !out_offsets || (nnz > 0 && (!indices || !out_indices))) TTX_FAIL(TTX_EINVAL, "NULL input");
  hipLaunchKernelGGL(split0_expand_kernel, dim3((unsigned)((nb + kCT / 8 - 1) / (kCT / 8))), dim3(kCT), 0, (hipStream_t)stream, nb,
                     k, p_rest, indices, offsets, out_indices, out_offsets);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

size_t ttx_preprocess_workspace_bytes(int64_t nnz) {
  return align_up((size_t)nnz * 4) + align_up((size_t)(num_units(nnz) + 64) * 4) + 256;
}

int ttx_preprocess_indices_sync(int64_t nnz, const int64_t* colidx, int64_t nb,
                                const int64_t* offsets, int32_t num_tables, int32_t warmup,
                                int64_t H, const int64_t* hashtbl, const int32_t* cache_state,
                                int64_t* rowidx, int64_t* tableidx, int64_t* pcol, int64_t* prow,
                                int32_t* ploc, int32_t* num_tt_host, int32_t* partitioned_host,
                                void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  return ttx_preprocess_indices_sync_fused(nnz, colidx, nb, offsets, num_tables, warmup, H, hashtbl, cache_state,
                                           rowidx, tableidx, pcol, prow, ploc, num_tt_host, partitioned_host,
                                           nullptr, nullptr, workspace, workspace_bytes, stream);
}

int ttx_preprocess_indices_sync_fused(int64_t nnz, const int64_t* colidx, int64_t nb,
                                      const int64_t* offsets, int32_t num_tables, int32_t warmup,
                                      int64_t H, const int64_t* hashtbl, const int32_t* cache_state,
                                      int64_t* rowidx, int64_t* tableidx, int64_t* pcol, int64_t* prow,
                                      int32_t* ploc, int32_t* num_tt_host, int32_t* partitioned_host,
                                      int64_t* upd_hashtbl, int64_t* upd_cache_freq,
                                      void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  return ttx_preprocess_indices_async(nnz, colidx, nb, offsets, num_tables, warmup, H, hashtbl, cache_state, rowidx,
                                      tableidx, pcol, prow, ploc, num_tt_host, partitioned_host, nullptr, upd_hashtbl,
                                      upd_cache_freq, workspace, workspace_bytes, stream);
}

int ttx_preprocess_indices_async(int64_t nnz, const int64_t* colidx, int64_t nb,
                                 const int64_t* offsets, int32_t num_tables, int32_t warmup,
                                 int64_t H, const int64_t* hashtbl, const int32_t* cache_state,
                                 int64_t* rowidx, int64_t* tableidx, int64_t* pcol, int64_t* prow,
                                 int32_t* ploc, int32_t* num_tt_host, int32_t* partitioned_host,
                                 int32_t* num_tt_dev, int64_t* upd_hashtbl, int64_t* upd_cache_freq,
                                 void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  return ttx_preprocess_indices_async_w(nnz, colidx, nb, offsets, num_tables, warmup, H, hashtbl, cache_state, rowidx,
                                        tableidx, pcol, prow, ploc, num_tt_host, partitioned_host, num_tt_dev,
                                        upd_hashtbl, upd_cache_freq, nullptr, nullptr, nullptr, workspace,
                                        workspace_bytes, stream);
}

int ttx_preprocess_indices_async_w(int64_t nnz, const int64_t* colidx, int64_t nb,
                                   const int64_t* offsets, int32_t num_tables, int32_t warmup,
                                   int64_t H, const int64_t* hashtbl, const int32_t* cache_state,
                                   int64_t* rowidx, int64_t* tableidx, int64_t* pcol, int64_t* prow,
                                   int32_t* ploc, int32_t* num_tt_host, int32_t* partitioned_host,
                                   int32_t* num_tt_dev, int64_t* upd_hashtbl, int64_t* upd_cache_freq,
                                   const float* psw, float* ppsw, int32_t* porig,
                                   void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (psw && !ppsw) TTX_FAIL(TTX_EINVAL, "per_sample_weights without a buffer for the partitioned weights");
  if (!num_tt_host || !partitioned_host) TTX_FAIL(TTX_EINVAL, "NULL output");
  *num_tt_host = (int32_t)nnz;
  *partitioned_host = 0;
  if (nnz == 0) return TTX_OK;  // cu:1389-1391
  if (nnz >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "nnz too large");
  if (num_tables <= 0 || nb % num_tables != 0)
    TTX_FAIL(TTX_EINVAL, "offsets has %lld bags, not a multiple of num_tables=%d", (long long)nb, num_tables);
  if (!colidx || !offsets || !rowidx || !tableidx) TTX_FAIL(TTX_EINVAL, "NULL input");
  const int32_t B = (int32_t)(nb / num_tables);
  const bool upd = upd_hashtbl && upd_cache_freq;
  const bool lookup = !(warmup || num_tables != 1);  // cu:1410-1412
  if ((upd || lookup) && (H <= 0 || H >= (1ll << 31)))
    TTX_FAIL(TTX_EINVAL, "hashtbl_size=%lld must be in (0, 2^31)", (long long)H);
  int32_t* loc = nullptr;
  int *unit_cnt = nullptr, *total = nullptr;
  const int N = (int)nnz, U = num_units(nnz);
  if (lookup) {
    if (!hashtbl || !cache_state || !pcol || !prow || !ploc) TTX_FAIL(TTX_EINVAL, "NULL cache input");
    if (!workspace || workspace_bytes < ttx_preprocess_workspace_bytes(nnz))
      TTX_FAIL(TTX_EWORKSPACE, "preprocess workspace too small");
    loc = (int32_t*)workspace;
    unit_cnt = (int*)((char*)workspace + align_up((size_t)nnz * 4));
    total = unit_cnt + U;
  }
  if (upd || lookup) {  // compute_rowidx + update_cache_state (cu:1077-1113) + cache_lookup in one launch
    const int64_t threads = nb * 8 > nnz ? nb * 8 : nnz;
    hipLaunchKernelGGL(rowidx_update_kernel, dim3((unsigned)((threads + kCT - 1) / kCT)), dim3(kCT), 0, st, nb, B,
                       offsets, rowidx, tableidx, nnz, colidx, (int32_t)H, upd ? upd_hashtbl : nullptr,
                       upd ? upd_cache_freq : nullptr, hashtbl, cache_state, loc, unit_cnt, ProBatch{}, 0ll);
  } else {
    hipLaunchKernelGGL(compute_rowidx_kernel, dim3((unsigned)((nb + kCT / 8 - 1) / (kCT / 8))), dim3(kCT), 0,
                       st, nb, B, offsets, rowidx, tableidx);
  }
  TTX_HIP(hipGetLastError());
  if (!lookup) {
    if (num_tt_dev) {  // every lookup is a TT lookup
      hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, st, num_tt_dev, (int32_t)nnz);
      TTX_HIP(hipGetLastError());
    }
    return TTX_OK;
  }
  const int scanned = U > kScanUnits ? 1 : 0;
  if (scanned) hipLaunchKernelGGL(scan_units_kernel, dim3(1), dim3(1024), 0, st, U, unit_cnt, total);
  hipLaunchKernelGGL(partition_scatter_kernel, dim3(U), dim3(kCT), 0, st, N, U, scanned, colidx, rowidx, loc,
                     unit_cnt, pcol, prow, ploc, total, num_tt_dev, ProBatch{}, 0ll, psw, ppsw, porig);
  TTX_HIP(hipGetLastError());
  *partitioned_host = 1;
  if (num_tt_dev) return TTX_OK;  // the split point stays on the device: no host synchronisation at all
                                  // (*num_tt_host keeps the upper bound nnz)
  // the one host synchronisation of the hot path (cu:1481-1488)
  TTX_HIP(hipMemcpyAsync(num_tt_host, total, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  TTX_HIP(hipStreamSynchronize(st));
  return TTX_OK;
}

size_t ttx_lookup_prologue_cached_multi_workspace_bytes(int32_t nbatch, int64_t nnz) {
  return (size_t)(nbatch > 0 ? nbatch : 0) * align_up(ttx_preprocess_workspace_bytes(nnz));
}

int ttx_lookup_prologue_cached_multi(const ttx_geom* g, int32_t nbatch, int64_t nnz, const int64_t* const* colidx_host,
                                     int64_t nb, const int64_t* const* offsets_host, int64_t H, int64_t* hashtbl,
                                     int64_t* cache_freq, const int32_t* cache_state, int64_t* rowidx,
                                     int64_t* tableidx, int64_t* pcol, int64_t* prow, int32_t* ploc,
                                     int32_t* num_tt_dev, void* plans, size_t plan_stride, void* workspace,
                                     size_t workspace_bytes, ttx_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  Dims d;
  int rc = make_dims(g, &d);
  if (rc != TTX_OK) return rc;
  if (nbatch <= 0 || nnz == 0) return TTX_OK;
  if (d.num_tables != 1) TTX_FAIL(TTX_EINVAL, "the cache serves one table (num_tables=%d)", d.num_tables);
  if (nnz >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "nnz too large");
  if (!colidx_host || !offsets_host || !hashtbl || !cache_freq || !cache_state || !rowidx || !tableidx || !pcol ||
      !prow || !ploc || !num_tt_dev || !plans)
    TTX_FAIL(TTX_EINVAL, "NULL input");
  if (nb <= 0) TTX_FAIL(TTX_EINVAL, "offsets must hold B + 1 entries");
  if (H <= 0 || H >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "hashtbl_size=%lld must be in (0, 2^31)", (long long)H);
  const size_t pb = plan_bytes(d, nnz);
  if (plan_stride < pb || plan_stride % 256 != 0)
    TTX_FAIL(TTX_EWORKSPACE, "plan stride %zu: need a multiple of 256 of at least %zu bytes", plan_stride, pb);
  const size_t wsb = align_up(ttx_preprocess_workspace_bytes(nnz));
  if (!workspace || workspace_bytes < (size_t)nbatch * wsb) TTX_FAIL(TTX_EWORKSPACE, "prologue workspace too small");
  for (int z = 0; z < nbatch; ++z)
    if (!colidx_host[z] || !offsets_host[z]) TTX_FAIL(TTX_EINVAL, "batch %d: NULL indices / offsets", z);
  const int N = (int)nnz, U = num_units(nnz);
  if (!plan_batches_ok(d, nnz) || U > kScanUnits) {  // general shape: batch after batch, same results
    for (int z = 0; z < nbatch; ++z) {
      int32_t n_host = 0, part = 0;
      rc = ttx_preprocess_indices_async(nnz, colidx_host[z], nb, offsets_host[z], 1, 0, H, hashtbl, cache_state,
                                        rowidx + (size_t)z * nnz, tableidx + (size_t)z * nnz, pcol + (size_t)z * nnz,
                                        prow + (size_t)z * nnz, ploc + (size_t)z * nnz, &n_host, &part, num_tt_dev + z,
                                        hashtbl, cache_freq, (char*)workspace + (size_t)z * wsb, wsb, stream);
      if (rc != TTX_OK) return rc;
      rc = plan_build(d, nnz, pcol + (size_t)z * nnz, tableidx + (size_t)z * nnz, prow + (size_t)z * nnz,
                      carve_plan(d, nnz, (char*)plans + (size_t)z * plan_stride), st, num_tt_dev + z);
      if (rc != TTX_OK) return rc;
    }
    return TTX_OK;
  }
  const long long ws_stride = (long long)(wsb / 4);
  for (int z0 = 0; z0 < nbatch; z0 += kMaxMulti) {  // kMaxMulti batches per launch (the pointer table of ProBatch)
    const int nz = nbatch - z0 < kMaxMulti ? nbatch - z0 : kMaxMulti;
    ProBatch mb{};
    for (int z = 0; z < nz; ++z) { mb.indices[z] = colidx_host[z0 + z]; mb.offsets[z] = offsets_host[z0 + z]; }
    mb.out_stride = nnz;
    mb.plan_stride = (long long)plan_stride;
    int32_t* loc = (int32_t*)((char*)workspace + (size_t)z0 * wsb);
    int* unit_cnt = (int*)((char*)loc + align_up((size_t)nnz * 4));
    int* total = unit_cnt + U;
    const int64_t threads = nb * 8 > nnz ? nb * 8 : nnz;
    hipLaunchKernelGGL(rowidx_update_kernel, dim3((unsigned)((threads + kCT - 1) / kCT), nz), dim3(kCT), 0, st, nb,
                       (int32_t)nb, offsets_host[z0], rowidx + (size_t)z0 * nnz, tableidx + (size_t)z0 * nnz, nnz,
                       colidx_host[z0], (int32_t)H, hashtbl, cache_freq, hashtbl, cache_state, loc, unit_cnt, mb,
                       ws_stride);
    TTX_HIP(hipGetLastError());
    hipLaunchKernelGGL(partition_scatter_kernel, dim3(U, nz), dim3(kCT), 0, st, N, U, 0, colidx_host[z0],
                       rowidx + (size_t)z0 * nnz, loc, unit_cnt, pcol + (size_t)z0 * nnz, prow + (size_t)z0 * nnz,
                       ploc + (size_t)z0 * nnz, total, num_tt_dev + z0, mb, ws_stride, (const float*)nullptr,
                       (float*)nullptr, (int32_t*)nullptr);
    TTX_HIP(hipGetLastError());
  }
  return plan_build_batches(d, nbatch, nnz, num_tt_dev, pcol, tableidx, prow, plans, plan_stride, st);
}

// (A/B knob of scripts/bench_cache.py: 1 = the 32-lane-group kernel for every D)
static TTX_KNOB(int, g_cache_fwd_lookup_groups, 0);
#ifdef TTX_TEST_HOOKS
int ttx_debug_cache_fwd(int32_t lookup_groups) { g_cache_fwd_lookup_groups = lookup_groups; return TTX_OK; }
#endif
// (for ttx_debug_state, ttx_tt.hip: bit 1 = the cache forward's A/B knob; bit 0 was the process-wide reference-exact switch of
//  rounds 3-5, now the per-call `flags` of ttx_cache_populate_f)
int ttx_cache_debug_state(void) { return g_cache_fwd_lookup_groups ? 2 : 0; }

int ttx_cache_forward(int32_t B, int64_t nnz, const int32_t* loc, const int64_t* rowidx, int32_t D,
                      const float* cache_weight, float* output, ttx_stream_t stream) {
  return ttx_cache_forward_n(B, nnz, nullptr, loc, rowidx, D, cache_weight, output, stream);
}

int ttx_cache_forward_n(int32_t B, int64_t nnz, const int32_t* skip_dev, const int32_t* loc,
                        const int64_t* rowidx, int32_t D, const float* cache_weight, float* output,
                        ttx_stream_t stream) {
  if (B <= 0) TTX_FAIL(TTX_EINVAL, "B=%d must be > 0", B);  // cu:1549
  if (D <= 0) TTX_FAIL(TTX_EINVAL, "D=%d must be > 0", D);
  if (nnz == 0) return TTX_OK;
  if (!loc || !rowidx || !cache_weight || !output) TTX_FAIL(TTX_EINVAL, "NULL input");
  ProfScope ps(TTX_PROF_CACHE_FWD, (hipStream_t)stream);
  if (D % 4 == 0 && ((((uintptr_t)cache_weight) | ((uintptr_t)output)) & 15) == 0 && !g_cache_fwd_lookup_groups)
    hipLaunchKernelGGL(cache_forward4_kernel, dim3((unsigned)((nnz + kCT - 1) / kCT)), dim3(kCT), 0, (hipStream_t)stream,
                       (int)nnz, D / 4, skip_dev, rowidx, loc, (const float4*)cache_weight, (float4*)output);
  else
    hipLaunchKernelGGL(cache_forward_kernel, dim3((unsigned)((nnz + kCT / 32 - 1) / (kCT / 32))), dim3(kCT), 0,
                       (hipStream_t)stream, (int)nnz, D, skip_dev, rowidx, loc, cache_weight, output);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

int ttx_cache_forward_nw(int32_t B, int64_t nnz, const int32_t* skip_dev, const int32_t* loc, const int64_t* rowidx,
                         const float* psw, int32_t D, const float* cache_weight, float* output, ttx_stream_t stream) {
  if (!psw) return ttx_cache_forward_n(B, nnz, skip_dev, loc, rowidx, D, cache_weight, output, stream);
  if (B <= 0) TTX_FAIL(TTX_EINVAL, "B=%d must be > 0", B);
  if (D <= 0) TTX_FAIL(TTX_EINVAL, "D=%d must be > 0", D);
  if (nnz == 0) return TTX_OK;
  if (!loc || !rowidx || !cache_weight || !output) TTX_FAIL(TTX_EINVAL, "NULL input");
  ProfScope ps(TTX_PROF_CACHE_FWD, (hipStream_t)stream);
  hipLaunchKernelGGL(cache_forward_w_kernel, dim3((unsigned)((nnz + kCT / 32 - 1) / (kCT / 32))), dim3(kCT), 0,
                     (hipStream_t)stream, (int)nnz, D, skip_dev, rowidx, loc, psw, cache_weight, output);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

int ttx_cache_rows_n(int64_t nnz, const int32_t* skip_dev, const int32_t* loc, int32_t D, const float* cache_weight,
                     float* rows, ttx_stream_t stream) {
  if (nnz == 0) return TTX_OK;
  if (D <= 0) TTX_FAIL(TTX_EINVAL, "D=%d must be > 0", D);
  if (!loc || !cache_weight || !rows) TTX_FAIL(TTX_EINVAL, "NULL input");
  hipLaunchKernelGGL(cache_rows_gather_kernel, dim3((unsigned)((nnz + kCT / 32 - 1) / (kCT / 32))), dim3(kCT), 0,
                     (hipStream_t)stream, (int)nnz, D, skip_dev, loc, cache_weight, rows);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

int ttx_cache_weighted_grad_n(int64_t nnz, const int32_t* skip_dev, int32_t D, const float* grad_output,
                              const int64_t* rowidx, const float* psw, float* scaled, int64_t* iota,
                              ttx_stream_t stream) {
  if (nnz == 0) return TTX_OK;
  if (D <= 0) TTX_FAIL(TTX_EINVAL, "D=%d must be > 0", D);
  if (!grad_output || !rowidx || !psw || !scaled || !iota) TTX_FAIL(TTX_EINVAL, "NULL input");
  hipLaunchKernelGGL(cache_scale_grad_kernel, dim3((unsigned)((nnz + kCT / 32 - 1) / (kCT / 32))), dim3(kCT), 0,
                     (hipStream_t)stream, (int)nnz, D, skip_dev, grad_output, rowidx, psw, scaled, iota);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

int ttx_cache_backward_sgd(int64_t nnz, int32_t D, const float* grad, const int32_t* loc,
                           const int64_t* rowidx, float lr, float* cache_weight, ttx_stream_t stream) {
  return ttx_cache_backward_sgd_n(nnz, nullptr, D, grad, loc, rowidx, lr, cache_weight, stream);
}

int ttx_cache_backward_sgd_n(int64_t nnz, const int32_t* skip_dev, int32_t D, const float* grad, const int32_t* loc,
                             const int64_t* rowidx, float lr, float* cache_weight, ttx_stream_t stream) {
  if (nnz == 0) return TTX_OK;
  if (D <= 0) TTX_FAIL(TTX_EINVAL, "D=%d must be > 0", D);
  if (!grad || !loc || !rowidx || !cache_weight) TTX_FAIL(TTX_EINVAL, "NULL input");
  launch_scatter_add(nnz, D, -lr, skip_dev, grad, loc, rowidx, cache_weight, (hipStream_t)stream);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

int ttx_cache_backward_dense(int64_t nnz, int32_t D, const float* grad, const int32_t* loc,
                             const int64_t* rowidx, int64_t cache_size, float* gcw, ttx_stream_t stream) {
  return ttx_cache_backward_dense_n(nnz, nullptr, D, grad, loc, rowidx, cache_size, gcw, stream);
}

int ttx_cache_backward_dense_n(int64_t nnz, const int32_t* skip_dev, int32_t D, const float* grad, const int32_t* loc,
                               const int64_t* rowidx, int64_t cache_size, float* gcw, ttx_stream_t stream) {
  if (D <= 0 || cache_size < 0) TTX_FAIL(TTX_EINVAL, "bad D / cache_size");
  if (!gcw) TTX_FAIL(TTX_EINVAL, "NULL output");
  if (cache_size > 0)
    TTX_HIP(hipMemsetAsync(gcw, 0, (size_t)cache_size * D * sizeof(float), (hipStream_t)stream));
  if (nnz == 0) return TTX_OK;
  if (!grad || !loc || !rowidx) TTX_FAIL(TTX_EINVAL, "NULL input");
  launch_scatter_add(nnz, D, 1.0f, skip_dev, grad, loc, rowidx, gcw, (hipStream_t)stream);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

int ttx_cache_backward_rowwise_adagrad_approx(int64_t nnz, int32_t D, const float* grad,
                                              const int32_t* loc, const int64_t* rowidx, float lr,
                                              float eps, float* state, float* cache_weight,
                                              ttx_stream_t stream) {
  return ttx_cache_backward_rowwise_adagrad_approx_n(nnz, nullptr, D, grad, loc, rowidx, lr, eps, state, cache_weight,
                                                     stream);
}

int ttx_cache_backward_rowwise_adagrad_approx_n(int64_t nnz, const int32_t* skip_dev, int32_t D, const float* grad,
                                                const int32_t* loc, const int64_t* rowidx, float lr, float eps,
                                                float* state, float* cache_weight, ttx_stream_t stream) {
  if (nnz == 0) return TTX_OK;
  if (D <= 0) TTX_FAIL(TTX_EINVAL, "D=%d must be > 0", D);
  if (!grad || !loc || !rowidx || !state || !cache_weight) TTX_FAIL(TTX_EINVAL, "NULL input");
  const int nmain = (int)((nnz + kCT / kWave - 1) / (kCT / kWave));
  const int K = (((uintptr_t)grad & 15) == 0) ? hot_rows(nnz, D, true) : 0;
  const int nhot = K * (int)((nnz + kHotSeg - 1) / kHotSeg);
  hipLaunchKernelGGL(cache_rowwise_adagrad_kernel, dim3((unsigned)(nmain + nhot)), dim3(kCT), 0, (hipStream_t)stream,
                     (int)nnz, D, skip_dev, grad, loc, rowidx, lr, eps, state, cache_weight, nmain, K);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

}  // extern "C"
namespace ttx {
// ---- the cache rows' update WITHOUT atomics (round 6): sorted, one writer per row, bit-identical from run to run ------------------
// The reference adds every cached lookup's bag gradient to its cache row with float atomics (cu:1574-1657, 1659-1733) and lets the
// row-wise Adagrad state race (cu:1735-1795); the kernels above do the same, so the order of additions -- and with it the last bits
// of cache_weight -- changes from run to run.  Here the batch's cached lookups are first grouped by cache row: the duplicate map of
// ttx_plan.hip (stable 64-bit radix sort + run heads: distinct rows ascending, a row's lookups in INDEX order) over the keys
// `cache row` (lookups in front of the split point get the key cache_size and form a last run nobody applies).  Then
//   SGD / dense:   Gu[u] = sum of the bag gradients of row u's lookups, in index order (gsum_slice / gsum_fold: every 16-lane group
//                  the same amount of work whatever the skew, no atomics), and ONE thread group per distinct row applies it;
//   row-wise Adagrad: the reference's per-lookup sequence old = state; state += g2; mult = lr / (sqrt(old + g2) + eps); w -= g mult,
//                  taken in index order within a row -- the order of the sequential oracle, oracle/ttx_oracle.c:499: a segmented
//                  inclusive scan of the lookups' g2 over the sorted order gives every lookup its multiplier, the same weighted sum
//                  Gu[u] = sum mult_n g_n follows, the row's last lookup leaves the new state.
// One pass over the gradient rows like the atomic kernels, plus the sort (16 B per lookup and 8-bit pass) and Gu (2 x 4 D per DISTINCT row).
__global__ __launch_bounds__(kCT) void cs_keys_kernel(int N, const int* __restrict__ skip_dev, const int32_t* __restrict__ loc,
                                                     long long cache_size, int64_t* __restrict__ keys, int64_t* __restrict__ vals) {
  const int i = blockIdx.x * kCT + threadIdx.x;
  if (i >= N) return;
  const int skip = skip_dev ? max(0, min(N, *skip_dev)) : 0;
  long long row = cache_size;  // (not a cached lookup: behind every real row)
  if (i >= skip) { const long long c = loc[i]; if (c >= 0 && c < cache_size) row = c; }
  keys[i] = cache_size - row;  // complemented: the descending sort leaves the rows ascending (all = cache_size + 1)
  vals[i] = i;
}

// one 16-lane group per distinct row (float4 columns; D % 4 != 0: float columns): dst[row] = dst[row] + scale Gu[u] (SGD: scale = -lr,
// Adagrad: -1 on the weighted sum) or dst[row] = Gu[u] (dense, onto the zeroed gradient)
template <typename V>
__global__ __launch_bounds__(kCT) void cs_apply_kernel(DedupMap M, int DV, long long cache_size, const V* __restrict__ Gu,
                                                      float scale, int assign, V* __restrict__ dst) {
  const int u = blockIdx.x * (kCT / 16) + threadIdx.x / 16, l = threadIdx.x & 15;
  if (u >= M.nu[0]) return;
  const long long row = M.uidx[u];
  if (row >= cache_size) return;  // the run of the lookups that are not cached
  const V* g = Gu + (size_t)u * DV;
  V* w = dst + (size_t)row * DV;
  for (int e = l; e < DV; e += 16) {
    if (assign) w[e] = g[e];
    else {
      V x = w[e];
      if constexpr (sizeof(V) == 16) { x.x += scale * g[e].x; x.y += scale * g[e].y; x.z += scale * g[e].z; x.w += scale * g[e].w; }
      else x += scale * g[e];
      w[e] = x;
    }
  }
}

// g2[b] = mean of the squares of bag gradient b (cu:1764-1772), a wave per bag, lanes striding the row, one butterfly
__global__ __launch_bounds__(kCT) void cs_bag_g2_kernel(int B, int D, const float* __restrict__ grad, float* __restrict__ g2) {
  const int b = blockIdx.x * (kCT / kWave) + threadIdx.x / kWave, lane = lane_id();
  if (b >= B) return;
  float s = 0.f;
  for (int e = lane; e < D; e += kWave) { const float x = grad[(size_t)b * D + e]; s = fmaf(x, x, s); }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
  if (lane == 0) g2[b] = s / D;
}

// Segmented inclusive scan of v_i = g2[bag of the lookup at sorted position i] over the positions, segments = the map's runs, in
// three steps over blocks of 1024 positions: (1) a block's TAIL = the sum of the positions behind its last run head (the whole
// block if it holds none) and whether it holds a head; (2) one work-group walks the blocks: carry into block b = tail of b - 1
// (+ the carry into b - 1 if that block held no head); (3) every position's prefix = carry (if no head lies in front of it in its
// block) + the in-block segmented scan.  Fixed association, no atomics.
constexpr int kCsBlock = 1024;
__device__ __forceinline__ float cs_seg_scan(float v, bool head, float* wsum, int* whead, float* tail, int* has_head,
                                             bool* no_head_before) {
  // -> inclusive segmented scan of v within the 1024-position block; *tail / *has_head: see above; *no_head_before: no head at or in
  // front of this position inside the block (the carry from the blocks before applies)
  const int lane = lane_id(), w = threadIdx.x / kWave;
  const unsigned long long hm = __ballot(head);
  // in-wave: sum of the lanes from the last head at or below this lane (or lane 0) up to this lane
  const unsigned long long below = hm & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
  const int start = below ? 63 - __clzll(below) : 0;  // first lane of this lane's run inside the wave
  float x = v;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const float y = __shfl_up(x, o, kWave);
    if (lane - o >= start) x += y;
  }
  // the wave's tail (sum from its last head, or the whole wave) and whether it holds a head
  const float wave_tail = __shfl(x, kWave - 1, kWave);  // (lane 63's run starts at the wave's last head: x[63] IS the tail)
  if (lane == 0) { wsum[w] = wave_tail; whead[w] = hm ? 1 : 0; }
  __syncthreads();
  // carry into this wave from the waves before it in the block: tails back to the nearest wave with a head
  float carry = 0.f;
  bool any = false;
  for (int k = w - 1; k >= 0 && !any; --k) { carry += wsum[k]; any = whead[k] != 0; }
  // NOTE the association: carry = wsum[w-1] + wsum[w-2] + ... (nearest first), fixed for a given batch
  const bool mine_no_head = below == 0;  // no head at or in front of this lane inside its wave
  if (mine_no_head) x += carry;
  *no_head_before = mine_no_head && !any;
  // the block's tail: from the last wave backwards to the nearest wave with a head
  float t = 0.f;
  bool h = false;
  for (int k = kCsBlock / kWave - 1; k >= 0 && !h; --k) { t += wsum[k]; h = whead[k] != 0; }
  *tail = t;
  *has_head = h ? 1 : 0;
  __syncthreads();
  return x;
}

__device__ __forceinline__ float cs_position_value(const DedupMap& M, int N, int i, const int64_t* __restrict__ rowidx,
                                                   const float* __restrict__ g2, bool* head, int* n_out) {
  *head = false;
  *n_out = -1;
  if (i >= N) return 0.f;
  const int n = M.occ[i];
  *n_out = n;
  *head = i == 0 || M.uid[M.occ[i - 1]] != M.uid[n];
  return g2[rowidx[n]];
}

__global__ __launch_bounds__(kCsBlock) void cs_scan_tails_kernel(DedupMap M, int N, const int64_t* __restrict__ rowidx,
                                                                const float* __restrict__ g2, float* __restrict__ blk_tail,
                                                                int* __restrict__ blk_head) {
  __shared__ float wsum[kCsBlock / kWave];
  __shared__ int whead[kCsBlock / kWave];
  bool head, nhb;
  int n;
  const float v = cs_position_value(M, N, blockIdx.x * kCsBlock + threadIdx.x, rowidx, g2, &head, &n);
  float tail;
  int hh;
  cs_seg_scan(v, head, wsum, whead, &tail, &hh, &nhb);
  if (threadIdx.x == 0) { blk_tail[blockIdx.x] = tail; blk_head[blockIdx.x] = hh; }
}

__global__ __launch_bounds__(1024) void cs_scan_carry_kernel(int nblk, const float* __restrict__ blk_tail, const int* __restrict__ blk_head,
                                                            float* __restrict__ blk_carry) {
  // one work-group: 1024 blocks at a time through LDS (coalesced in, coalesced out), ONE lane walks them in order -- a fixed
  // association; the walk is a thousand LDS round trips per million lookups (first form: one lane on global memory, 72 us at 1 M)
  __shared__ float tl[1024], cr[1024];
  __shared__ int hd[1024];
  __shared__ float c_run;
  if (threadIdx.x == 0) c_run = 0.f;
  for (int b0 = 0; b0 < nblk; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    if (b < nblk) { tl[threadIdx.x] = blk_tail[b]; hd[threadIdx.x] = blk_head[b]; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float c = c_run;
      const int n = min(1024, nblk - b0);
      for (int k = 0; k < n; ++k) {
        cr[k] = c;
        c = hd[k] ? tl[k] : c + tl[k];
      }
      c_run = c;
    }
    __syncthreads();
    if (b < nblk) blk_carry[b] = cr[threadIdx.x];
    __syncthreads();
  }
}

// every cached lookup's multiplier (psw for the weighted sum) and, from a row's LAST lookup, the row's new state
__global__ __launch_bounds__(kCsBlock) void cs_scan_emit_kernel(DedupMap M, int N, long long cache_size, const int64_t* __restrict__ rowidx,
                                                               const float* __restrict__ g2, const float* __restrict__ blk_carry,
                                                               float lr, float eps, float* __restrict__ state, float* __restrict__ mult) {
  __shared__ float wsum[kCsBlock / kWave];
  __shared__ int whead[kCsBlock / kWave];
  const int i = blockIdx.x * kCsBlock + threadIdx.x;
  bool head, nhb;
  int n;
  const float v = cs_position_value(M, N, i, rowidx, g2, &head, &n);
  float tail;
  int hh;
  float incl = cs_seg_scan(v, head, wsum, whead, &tail, &hh, &nhb);
  if (i >= N) return;
  if (nhb) incl += blk_carry[blockIdx.x];
  const int u = M.uid[n];
  const long long row = M.uidx[u];
  if (row >= cache_size) { mult[n] = 0.f; return; }  // not a cached lookup
  const float s0 = state[row];  // (read by every lookup of the row, possibly in other blocks: the new state goes to a side array)
  mult[n] = (float)(lr * (1.0 / (sqrtf(s0 + incl) + eps)));  // old + g2 = state + inclusive prefix (cu:1781-1782, double intermediate)
  const bool last = i + 1 >= N || M.uid[M.occ[i + 1]] != u;
  if (last) ((float*)M.iota)[u] = s0 + incl;  // (M.iota is unused by this map: the rows' new states, applied by cs_state_kernel)
}
__global__ __launch_bounds__(kCT) void cs_state_kernel(DedupMap M, long long cache_size, float* __restrict__ state) {
  const int u = blockIdx.x * kCT + threadIdx.x;
  if (u >= M.nu[0]) return;
  const long long row = M.uidx[u];
  if (row < cache_size) state[row] = ((const float*)M.iota)[u];
}

// ---- the same update for SMALL batches, one launch (round 6): every work-group OWNS the cache rows `row % G == g` -----------------
// The sorted update above is a chain of ~16 launches (~65 us whatever the batch).  Up to kOwnMaxN lookups the grouping needs no
// global sort: work-group g scans the batch's cached lookups for ITS rows (a stable compaction into LDS, index order), groups them
// by row with a counting sort over its local row ids row / G (counts by LDS integer atomics -- order-free --, a scan, then ONE wave
// places the entries batch after batch so that equal rows keep their index order), and sums every row's bag gradients in that order:
// short runs by one 16-lane group each, long runs (a hot row: a sixth of a Zipf batch) by all 64 groups, strided, part sums folded
// in group order.  One owner per row: plain read-modify-write, no atomics, the same bits every run.  Row-wise Adagrad walks a row's
// lookups in index order (the sequential oracle's order); a long run gets its multipliers from a wave scan of the lookups' g2 first.
// Rows are numbered by descending frequency (cache_populate), so `row % G` spreads the hot rows over the work-groups.
constexpr int kOwnThreads = 1024, kOwnWaves = kOwnThreads / kWave, kOwnGroups = kOwnThreads / 16;
constexpr int kOwnCap = 2048;      // list entries a work-group takes per round (more: further rounds, in index order)
constexpr int kOwnLong = 48;       // a run beyond this many lookups is summed by the whole work-group
constexpr int kOwnMaxLR = 4096;    // local row ids per work-group (cache_size / G rounded up)
constexpr int kOwnMaxN = 32768;    // every work-group reads all N cache locations: beyond this the sorted update
constexpr int kOwnMaxD4 = 64;      // D <= 256, D % 4 == 0
struct OwnArgs {
  int N, D4, optim, G, gshift, LR, B;  // G = 1 << gshift work-groups
  float lr, eps;
  long long cache_size;
  const int* skip_dev;
  const float4* grad;
  const int32_t* loc;
  const int64_t* rowidx;
  float* state;
  float4* dst;
};
__device__ __forceinline__ float own_group_sum16(float v) {  // sum over the 16 lanes of a group (all lanes get it)
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
  return v;
}
__device__ __forceinline__ float own_dot(const float4& a) { return fmaf(a.x, a.x, fmaf(a.y, a.y, fmaf(a.z, a.z, a.w * a.w))); }
__device__ __forceinline__ void own_axpy(float4& acc, float m, const float4& v) {
  acc.x = fmaf(m, v.x, acc.x); acc.y = fmaf(m, v.y, acc.y); acc.z = fmaf(m, v.z, acc.z); acc.w = fmaf(m, v.w, acc.w);
}

__global__ __launch_bounds__(kOwnThreads) void cache_update_owner_kernel(OwnArgs a) {
  extern __shared__ __attribute__((aligned(16))) int own_lds[];
  int* l_row = own_lds;                 // [cap] list: cache row, index order
  int* l_bag = l_row + kOwnCap;         // [cap] its bag row
  int* s_row = l_bag + kOwnCap;         // [cap] the same grouped by row
  int* s_bag = s_row + kOwnCap;
  int* seg = s_bag + kOwnCap;           // [cap + 2] run starts (+ end)
  int* lng = seg + kOwnCap + 2;         // [cap / kOwnLong + 2] indices of the long runs
  int* cnt = lng + kOwnCap / kOwnLong + 2;  // [LR] counts -> starts -> running positions
  float* g2s = (float*)(cnt + a.LR);    // [cap] g2 of a long run's lookups, then their multipliers
  float4* fold = (float4*)(((uintptr_t)(g2s + kOwnCap) + 15) & ~(uintptr_t)15);  // [groups][D4] part sums of a long run
  __shared__ int wcnt[kOwnWaves];
  __shared__ int s_scan[kOwnWaves + 1];
  __shared__ int s_nseg, s_nlong;
  __shared__ float s_run;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
  const int grp = tid >> 4, gl = tid & 15;
  const int g = blockIdx.x, G = a.G, D4 = a.D4, N = a.N;
  const int nv = (D4 + 15) / 16;  // float4 columns per lane (<= 4)
  const int skip = a.skip_dev ? max(0, min(N, *a.skip_dev)) : 0;

  auto process = [&](int n) {  // (every thread calls it with the same n: barriers inside)
    // ---- group the list by row: counting sort over the local row ids, stable ----
    for (int e = tid; e < a.LR; e += kOwnThreads) cnt[e] = 0;
    for (int e = tid; e < n; e += kOwnThreads) l_bag[e] = (int)a.rowidx[l_bag[e]];  // position -> bag row
    __syncthreads();
    for (int e = tid; e < n; e += kOwnThreads) atomicAdd(&cnt[l_row[e] >> a.gshift], 1);
    __syncthreads();
    {  // exclusive scan of cnt[0, LR): a strip per thread
      const int per = (a.LR + kOwnThreads - 1) / kOwnThreads;
      const int b0 = tid * per, b1 = min(a.LR, b0 + per);
      int sum = 0;
      for (int e = b0; e < b1; ++e) sum += cnt[e];
      const int inc = wave_incl_scan(sum);
      if (lane == kWave - 1) s_scan[w] = inc;
      __syncthreads();
      int base = 0;
      for (int k = 0; k < w; ++k) base += s_scan[k];
      int run = base + inc - sum;
      for (int e = b0; e < b1; ++e) { const int c = cnt[e]; cnt[e] = run; run += c; }
    }
    __syncthreads();
    if (w == 0) {  // ONE wave places the entries, 64 at a time, in list order: equal rows keep their index order
      for (int b0 = 0; b0 < n; b0 += kWave) {
        const int e = b0 + lane;
        const bool valid = e < n;
        const int row = valid ? l_row[e] : -1, bag = valid ? l_bag[e] : 0;
        const int key = valid ? (row >> a.gshift) : -1 - lane;  // (invalid lanes: keys of their own)
        // rank among the batch's lanes with the same key, and how many there are: 64 lane broadcasts, no LDS, no serial turn per
        // distinct key (the first version took a turn -- an LDS round trip -- per distinct key: 64 per batch on a uniform stream)
        int before = 0, same = 0;
#pragma unroll
        for (int j = 0; j < kWave; ++j) {
          const int kj = __builtin_amdgcn_readlane(key, j);
          const int eq = kj == key ? 1 : 0;
          same += eq;
          before += (j < lane) ? eq : 0;
        }
        if (valid) {
          const int basep = cnt[key];
          const int dest = basep + before;
          s_row[dest] = row;
          s_bag[dest] = bag;
          if (before == same - 1) cnt[key] = basep + same;  // (the key's last lane of the batch: one writer per key)
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    // ---- run heads -> seg[], long runs -> lng[] ----
    if (tid == 0) { s_nseg = 0; s_nlong = 0; }
    __syncthreads();
    for (int b0 = 0; b0 < n; b0 += kOwnThreads) {  // (uniform)
      const int j = b0 + tid;
      const bool head = j < n && (j == 0 || s_row[j] != s_row[j - 1]);
      const unsigned long long hm = __ballot(head);
      if (lane == 0) wcnt[w] = __popcll(hm);
      __syncthreads();
      int off = 0, tot = 0;
      for (int k = 0; k < kOwnWaves; ++k) { const int c = wcnt[k]; if (k < w) off += c; tot += c; }
      const int base = s_nseg;
      if (head) seg[base + off + __popcll(hm & lanemask_lt())] = j;
      __syncthreads();
      if (tid == 0) s_nseg = base + tot;
      __syncthreads();
    }
    const int nseg = s_nseg;
    if (tid == 0) seg[nseg] = n;
    __syncthreads();
    for (int s0 = tid; s0 < nseg; s0 += kOwnThreads)
      if (seg[s0 + 1] - seg[s0] > kOwnLong) lng[atomicAdd(&s_nlong, 1)] = s0;  // (which long run first does not touch any value)
    __syncthreads();
    const int nlong = s_nlong;
    // ---- short runs: one 16-lane group each, lookup after lookup in index order ----
    for (int s0 = grp; s0 < nseg; s0 += kOwnGroups) {
      const int b = seg[s0], e = seg[s0 + 1];
      if (e - b > kOwnLong) continue;
      const int row = s_row[b];
      float4 acc[4], wv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        // (the row to update is requested now, with the first gradient rows -- not after the sums, one more trip to memory later)
        wv[k] = (k < nv && gl + 16 * k < D4) ? a.dst[(size_t)row * D4 + gl + 16 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float run = a.optim == TTX_OPTIM_ADAGRAD ? a.state[row] : 0.f;
      for (int j = b; j < e; j += 4) {
        float4 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int bag = j + u < e ? s_bag[j + u] : -1;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            v[u][k] = (bag >= 0 && k < nv && gl + 16 * k < D4) ? a.grad[(size_t)bag * D4 + gl + 16 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j + u >= e) break;  // (group-uniform)
          float m = 1.f;
          if (a.optim == TTX_OPTIM_ADAGRAD) {
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sq += own_dot(v[u][k]);
            const float g2 = own_group_sum16(sq) / (float)(4 * D4);
            m = (float)(a.lr * (1.0 / (sqrtf(run + g2) + a.eps)));  // (cu:1781-1782: double intermediate)
            run += g2;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) own_axpy(acc[k], m, v[u][k]);
        }
      }
      const float scale = a.optim == TTX_OPTIM_SGD ? -a.lr : (a.optim == TTX_OPTIM_ADAGRAD ? -1.f : 1.f);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nv && gl + 16 * k < D4) {
          // (dense: onto the zeroed gradient -- an add, not a store: a row whose lookups span two rounds of the list comes twice)
          float4 x = wv[k];
          own_axpy(x, scale, acc[k]);
          a.dst[(size_t)row * D4 + gl + 16 * k] = x;
        }
      if (a.optim == TTX_OPTIM_ADAGRAD && gl == 0) a.state[row] = run;
    }
    // ---- long runs: the whole work-group, strided; part sums folded in group order ----
    for (int h = 0; h < nlong; ++h) {  // (uniform)
      const int s0 = lng[h];
      const int b = seg[s0], e = seg[s0 + 1], len = e - b;
      const int row = s_row[b];
      if (a.optim == TTX_OPTIM_ADAGRAD) {
        for (int j = grp; j < len; j += kOwnGroups) {  // g2 of every lookup of the run
          const int bag = s_bag[b + j];
          float sq = 0.f;
          for (int k = 0; k < nv; ++k)
            if (gl + 16 * k < D4) sq += own_dot(a.grad[(size_t)bag * D4 + gl + 16 * k]);
          const float g2 = own_group_sum16(sq) / (float)(4 * D4);
          if (gl == 0) g2s[j] = g2;
        }
        __syncthreads();
        if (w == 0) {  // multipliers in index order: old + g2 = state + inclusive prefix
          float run = a.state[row];
          for (int j0 = 0; j0 < len; j0 += kWave) {
            const float v = j0 + lane < len ? g2s[j0 + lane] : 0.f;
            const float inc = wave_incl_scan_f(v);
            if (j0 + lane < len) g2s[j0 + lane] = (float)(a.lr * (1.0 / (sqrtf(run + inc) + a.eps)));
            run += __shfl(inc, kWave - 1, kWave);
          }
          if (lane == 0) s_run = run;
        }
        __syncthreads();
      }
      float4 acc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = grp; j < len; j += 4 * kOwnGroups) {
        float4 v[4][4];
        float m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jj = j + u * kOwnGroups;
          const int bag = jj < len ? s_bag[b + jj] : -1;
          m[u] = (jj < len && a.optim == TTX_OPTIM_ADAGRAD) ? g2s[jj] : 1.f;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            v[u][k] = (bag >= 0 && k < nv && gl + 16 * k < D4) ? a.grad[(size_t)bag * D4 + gl + 16 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j + u * kOwnGroups < len) {
#pragma unroll
            for (int k = 0; k < 4; ++k) own_axpy(acc[k], m[u], v[u][k]);
          }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nv && gl + 16 * k < D4) fold[(size_t)grp * D4 + gl + 16 * k] = acc[k];
      __syncthreads();
      if (tid < D4) {
        float4 t = fold[tid];
        for (int q = 1; q < kOwnGroups; ++q) {
          const float4 x = fold[(size_t)q * D4 + tid];
          t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
        }
        const float scale = a.optim == TTX_OPTIM_SGD ? -a.lr : (a.optim == TTX_OPTIM_ADAGRAD ? -1.f : 1.f);
        float4* wp = a.dst + (size_t)row * D4 + tid;
        float4 x = *wp;
        own_axpy(x, scale, t);
        *wp = x;
        if (a.optim == TTX_OPTIM_ADAGRAD && tid == 0) a.state[row] = s_run;
      }
      __syncthreads();
    }
  };

  // The scan: rounds of kOwnThreads positions; the cache location of the round after the next is requested before this round's
  // barriers (a load per round in front of its barrier was 15 of the first version's 65 us at 10k lookups).  The list is
  // processed when the next round might not fit -- ONE call site: the body is large, and a copy per round of an unrolled scan
  // (second version) ran out of the instruction cache: 150 us.
  const int nrounds = (N - skip + kOwnThreads - 1) / kOwnThreads;
  auto fetch = [&](int r) -> int {
    const int i = skip + r * kOwnThreads + tid;
    return (r < nrounds && i < N) ? a.loc[i] : -1;
  };
  int c0 = fetch(0), c1 = fetch(1);
  int r = 0;
  while (r < nrounds) {  // (uniform)
    int n = 0;
    for (; r < nrounds; ++r) {
      const int c2 = fetch(r + 2);
      const int i = skip + r * kOwnThreads + tid;
      const int c = c0;
      const bool m = c >= 0 && (long long)c < a.cache_size && (c & (G - 1)) == g;
      const unsigned long long bm = __ballot(m);
      if (lane == 0) wcnt[w] = __popcll(bm);
      __syncthreads();
      int off = 0, tot = 0;
      for (int k = 0; k < kOwnWaves; ++k) { const int cc = wcnt[k]; if (k < w) off += cc; tot += cc; }
      __syncthreads();  // (wcnt is rewritten by the next round)
      if (n + tot > kOwnCap) break;  // (uniform) the list is full: this round's rows are applied, the rest follows in a later turn
      if (m) {
        const int p = n + off + __popcll(bm & lanemask_lt());
        l_row[p] = c;
        l_bag[p] = i;  // (the lookup's position: process() turns it into the bag row, one round of loads for the whole list)
      }
      n += tot;
      c0 = c1; c1 = c2;
    }
    __syncthreads();
    if (n > 0) process(n);
    __syncthreads();
  }
}

static size_t own_lds_bytes(int LR, int D4) {
  return (size_t)(5 * kOwnCap + 2 + kOwnCap / kOwnLong + 2 + LR) * 4 + (size_t)kOwnCap * 4 + 16 + (size_t)kOwnGroups * D4 * 16;
}
// G work-groups for this cache size, or 0: the batch / shape is not this kernel's
static int own_groups(int64_t nnz, int64_t cache_size, int32_t D, const void* grad, const void* dst) {
  if (nnz > kOwnMaxN || D % 4 != 0 || D / 4 > kOwnMaxD4 || ((((uintptr_t)grad) | ((uintptr_t)dst)) & 15) != 0) return 0;
  int G = 128;
  while ((cache_size + G - 1) / G > kOwnMaxLR && G < 4096) G *= 2;
  if ((cache_size + G - 1) / G > kOwnMaxLR) return 0;
  return G;
}

static size_t cs_scan_bytes(int64_t nnz, int64_t B) {
  const size_t nblk = ((size_t)nnz + kCsBlock - 1) / kCsBlock;
  return align_up((size_t)(B > 0 ? B : 1) * 4) + 3 * align_up(nblk * 4) + align_up((size_t)nnz * 4);
}

}  // namespace ttx
extern "C" {

// test hook: that sort on its own (tests/test_primref_gpu.py checks it against hipCUB's SortPairsDescending, the
// library call the reference makes).  keys / vals are copied to keys_out / vals_out.
size_t ttx_debug_sort_workspace_bytes(int64_t n) { return sort_pairs_ws_bytes(n); }

size_t ttx_cache_backward_sorted_workspace_bytes(int64_t nnz, int64_t num_bags, int32_t D) {
  if (nnz <= 0 || D <= 0 || num_bags < 0) return 0;
  return align_up(dedup_bytes(nnz)) + align_up((size_t)(nnz + 1) * D * sizeof(float)) + gsum_scratch_bytes(D, nnz) +
         cs_scan_bytes(nnz, num_bags);
}

int ttx_cache_backward_sorted(int32_t optim, int64_t nnz, const int32_t* skip_dev, int64_t num_bags, int32_t D, const float* grad,
                              const int32_t* loc, const int64_t* rowidx, float lr, float eps, int64_t cache_size,
                              float* cache_optimizer_state, float* dst, void* workspace, size_t workspace_bytes,
                              ttx_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (optim != TTX_OPTIM_SGD && optim != TTX_OPTIM_ADAGRAD && optim != TTX_OPTIM_DENSE) TTX_FAIL(TTX_EINVAL, "unknown optimizer %d", optim);
  if (D <= 0 || cache_size < 0 || cache_size >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "bad D / cache_size");
  if (!dst) TTX_FAIL(TTX_EINVAL, "NULL output");
  if (optim == TTX_OPTIM_DENSE && cache_size > 0)
    TTX_HIP(hipMemsetAsync(dst, 0, (size_t)cache_size * D * sizeof(float), st));
  if (nnz == 0 || cache_size == 0) return TTX_OK;
  if (nnz >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "nnz=%lld out of range", (long long)nnz);
  if (!grad || !loc || !rowidx) TTX_FAIL(TTX_EINVAL, "NULL input");
  if (optim == TTX_OPTIM_ADAGRAD && (!cache_optimizer_state || num_bags <= 0))
    TTX_FAIL(TTX_EINVAL, "row-wise Adagrad needs cache_optimizer_state and the number of bags");
  if (const int G = own_groups(nnz, cache_size, D, grad, dst)) {  // small batch: one launch, every work-group owns its rows
    OwnArgs A;
    A.N = (int)nnz; A.D4 = D / 4; A.optim = optim; A.G = G; A.LR = (int)((cache_size + G - 1) / G); A.B = (int)num_bags;
    A.gshift = 0;
    while ((1 << A.gshift) < G) ++A.gshift;
    A.lr = lr; A.eps = eps; A.cache_size = cache_size; A.skip_dev = skip_dev; A.grad = (const float4*)grad; A.loc = loc;
    A.rowidx = rowidx; A.state = cache_optimizer_state; A.dst = (float4*)dst;
    const size_t lds = own_lds_bytes(A.LR, A.D4);
    const int rc_attr = allow_dynamic_lds((const void*)cache_update_owner_kernel, (int)lds);
    if (rc_attr) return rc_attr;
    hipLaunchKernelGGL(cache_update_owner_kernel, dim3((unsigned)G), dim3(kOwnThreads), lds, st, A);
    TTX_HIP(hipGetLastError());
    return TTX_OK;
  }
  if (!workspace || workspace_bytes < ttx_cache_backward_sorted_workspace_bytes(nnz, num_bags, D))
    TTX_FAIL(TTX_EWORKSPACE, "sorted cache update: workspace too small");
  char* ws = (char*)workspace;
  const DedupMap M = carve_dedup(nnz, ws);
  float* Gu = (float*)(ws + align_up(dedup_bytes(nnz)));
  char* gscr = (char*)Gu + align_up((size_t)(nnz + 1) * D * sizeof(float));
  char* sc = gscr + gsum_scratch_bytes(D, nnz);
  const int N = (int)nnz;
  int64_t *keys, *vals;
  dedup_key_buffers(M, nnz, &keys, &vals);
  hipLaunchKernelGGL(cs_keys_kernel, dim3((unsigned)((N + kCT - 1) / kCT)), dim3(kCT), 0, st, N, skip_dev, loc, (long long)cache_size,
                     keys, vals);
  int rc = dedup_build_from_keys(nnz, (unsigned long long)cache_size + 1ull, M, st);
  if (rc) return rc;
  const float* psw = nullptr;
  if (optim == TTX_OPTIM_ADAGRAD) {
    const int nblk = (N + kCsBlock - 1) / kCsBlock;
    float* g2 = (float*)sc;
    float* blk_tail = (float*)(sc + align_up((size_t)num_bags * 4));
    int* blk_head = (int*)((char*)blk_tail + align_up((size_t)nblk * 4));
    float* blk_carry = (float*)((char*)blk_head + align_up((size_t)nblk * 4));
    float* mult = (float*)((char*)blk_carry + align_up((size_t)nblk * 4));
    hipLaunchKernelGGL(cs_bag_g2_kernel, dim3((unsigned)((num_bags + kCT / kWave - 1) / (kCT / kWave))), dim3(kCT), 0, st, (int)num_bags,
                       D, grad, g2);
    hipLaunchKernelGGL(cs_scan_tails_kernel, dim3(nblk), dim3(kCsBlock), 0, st, M, N, rowidx, g2, blk_tail, blk_head);
    hipLaunchKernelGGL(cs_scan_carry_kernel, dim3(1), dim3(1024), 0, st, nblk, blk_tail, blk_head, blk_carry);
    hipLaunchKernelGGL(cs_scan_emit_kernel, dim3(nblk), dim3(kCsBlock), 0, st, M, N, (long long)cache_size, rowidx, g2, blk_carry, lr, eps,
                       cache_optimizer_state, mult);
    hipLaunchKernelGGL(cs_state_kernel, dim3((unsigned)((N + kCT - 1) / kCT)), dim3(kCT), 0, st, M, (long long)cache_size,
                       cache_optimizer_state);
    TTX_HIP(hipGetLastError());
    psw = mult;
  }
  rc = gsum_launch(M, nnz, 0, D, rowidx, nullptr, psw, grad, Gu, gscr, st);
  if (rc) return rc;
  const float scale = optim == TTX_OPTIM_SGD ? -lr : -1.0f;
  const unsigned groups = (unsigned)((N + kCT / 16 - 1) / (kCT / 16));  // (nu <= N lives on the device: sized by N, surplus groups leave)
  if (D % 4 == 0 && ((((uintptr_t)dst) | ((uintptr_t)Gu)) & 15) == 0)
    hipLaunchKernelGGL(cs_apply_kernel<float4>, dim3(groups), dim3(kCT), 0, st, M, D / 4, (long long)cache_size, (const float4*)Gu, scale,
                       optim == TTX_OPTIM_DENSE ? 1 : 0, (float4*)dst);
  else
    hipLaunchKernelGGL(cs_apply_kernel<float>, dim3(groups), dim3(kCT), 0, st, M, D, (long long)cache_size, (const float*)Gu, scale,
                       optim == TTX_OPTIM_DENSE ? 1 : 0, dst);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

int ttx_debug_sort_pairs_desc(int64_t n, const int64_t* keys, const int64_t* vals, int64_t* keys_out, int64_t* vals_out,
                              void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  (void)hipGetLastError();
  if (n <= 0 || n >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "n=%lld out of range", (long long)n);
  if (!keys || !vals || !keys_out || !vals_out) TTX_FAIL(TTX_EINVAL, "NULL input");
  if (!workspace || workspace_bytes < ttx_debug_sort_workspace_bytes(n)) TTX_FAIL(TTX_EWORKSPACE, "sort workspace too small");
  int64_t *sk = nullptr, *sv = nullptr;
  const int rc = sort_pairs_desc(n, keys, vals, (char*)workspace, &sk, &sv, (hipStream_t)stream, -1);
  if (rc) return rc;
  TTX_HIP(hipMemcpyAsync(keys_out, sk, (size_t)n * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  TTX_HIP(hipMemcpyAsync(vals_out, sv, (size_t)n * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return TTX_OK;
}

size_t ttx_cache_populate_workspace_bytes(const ttx_geom* g, int64_t H, int64_t cache_size, int32_t D) {
  (void)D;
  Dims d;
  if (make_dims(g, &d) != TTX_OK || H <= 0 || cache_size < 0) return 0;
  int WT, U;
  unit_shape(H, &WT, &U);
  return 4 * align_up((size_t)H * 8) + align_up((size_t)256 * U * 4) + 2048 + plan_bytes(d, cache_size) + 256;
}

int ttx_cache_populate(const ttx_geom* g, const float* const* tt_cores, int64_t H, int64_t* hashtbl,
                       int64_t* cache_freq, int32_t* cache_state, int64_t cache_size, int32_t D,
                       float* cache_weight, void* workspace, size_t workspace_bytes,
                       ttx_stream_t stream) {
  return ttx_cache_populate_f(g, tt_cores, H, hashtbl, cache_freq, cache_state, cache_size, D, cache_weight, 0, workspace,
                              workspace_bytes, stream);
}

// flags bit 0 (TTX_POPULATE_REFERENCE_EXACT): leave the cache_state of an evicted slot as it was, like the reference's
// mark_popular_colidx_kernel (tt_embeddings_cuda.cu:1131-1133) -- the deliberate fix switched off, per call.
int ttx_cache_populate_f(const ttx_geom* g, const float* const* tt_cores, int64_t H, int64_t* hashtbl,
                         int64_t* cache_freq, int32_t* cache_state, int64_t cache_size, int32_t D,
                         float* cache_weight, int32_t flags, void* workspace, size_t workspace_bytes,
                         ttx_stream_t stream) {
  if (flags & ~TTX_POPULATE_REFERENCE_EXACT) TTX_FAIL(TTX_EINVAL, "unknown cache_populate flags %d", flags);
  hipStream_t st = (hipStream_t)stream;
  Dims d;
  int rc = make_dims(g, &d);
  if (rc) return rc;
  if (d.tab || d.num_tables != 1) TTX_FAIL(TTX_EINVAL, "cache_populate serves one table (tt_embeddings_ops.py:456)");
  // cu:1271-1274
  if (H <= 0 || H >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "hashtbl_size=%lld must be in (0, 2^31)", (long long)H);
  if (cache_size < 0 || cache_size > H) TTX_FAIL(TTX_EINVAL, "cache_size=%lld must be <= hashtbl_size", (long long)cache_size);
  if (!hashtbl || !cache_freq || !cache_state || !tt_cores) TTX_FAIL(TTX_EINVAL, "NULL input");
  if (!workspace || workspace_bytes < ttx_cache_populate_workspace_bytes(g, H, cache_size, D))
    TTX_FAIL(TTX_EWORKSPACE, "cache_populate workspace too small");
  char* ws = (char*)workspace;
  int64_t* sorted_keys = nullptr;
  rc = sort_pairs_desc(H, cache_freq, hashtbl, ws, nullptr, &sorted_keys, st, -1);
  if (rc) return rc;
  int WT, U;
  unit_shape(H, &WT, &U);
  char* rows_ws = ws + 4 * align_up((size_t)H * 8) + align_up((size_t)256 * U * 4) + 2048;
  hipLaunchKernelGGL(mark_popular_kernel, dim3((unsigned)((H + kCT - 1) / kCT)), dim3(kCT), 0, st, (int32_t)H,
                     cache_size, sorted_keys, hashtbl, cache_freq, cache_state, flags & TTX_POPULATE_REFERENCE_EXACT);
  TTX_HIP(hipGetLastError());
  if (cache_size == 0) return TTX_OK;
  if (!cache_weight) TTX_FAIL(TTX_EINVAL, "cache_weight is NULL");
  return ttx_tt_rows(g, D, cache_size, sorted_keys, nullptr, tt_cores, cache_weight, rows_ws,
                     plan_bytes(d, cache_size), stream);
}

}  // extern "C"
