// ttx_plan.hip -- lookup plan: index decode + stable radix sorts by core slice.
//
// Replaces the reference's per-chunk set-up kernels
// (init_batch_gemm_{forward,backward}_*T_kernel, tt_embeddings_cuda.cu:79-360,
// :754-918) and the need for float atomics in update_d_tt_cores_kernel
// (:362-377): after this kernel every core slice knows, in index order, the
// lookups that touch it, so gradients are reduced by one owner per slice.
//
// What a plan holds, per core t:
//   1. the slice id of every lookup, sid = table*p_t + i_t, with i_t decoded exactly as
//      the reference does (idx / L_t % p_t, cu:795-799; when prod(p) <= 2^32 the two int64
//      divisions become 32-bit magic-number multiplies);
//   2. a stable LSD radix sort (8-bit digits) of the lookups by sid.  Ranking inside a wave
//      uses 9 ballots per 64 keys (wave_match8); no float or ordering-dependent atomics, so
//      the order -- and every sum that follows it -- is deterministic;
//   3. slice offsets (thin cores) / the chunk work-list (pivot core 1);
//   4. (pivot only) one flat record per sorted lookup {n, sid_0, sid_2, sid_3} and its bag
//      row, so a contraction wave reaches its operands with two dependent loads.
// Three routes, all producing the same plan:
//   * mb_single_kernel  -- every sort is one 8-bit pass and nnz <= 16384 (the benchmark
//     shape): ONE launch; each work-group histograms its core's keys itself in LDS, then
//     ranks and scatters its own 256 positions; work-group 0 of each core also emits the
//     offset table / chunk list from the digit totals.
//   * mb_count / (mb_scan) / mb_scatter per pass + mb_finish -- any size, any slice count;
//     the scan launch is folded into the scatter pass while there are <= 96 wave units and
//     the finish launch into it when every sort is a single pass.
//   * plan_small_kernel -- nnz <= 1024: one work-group per core, keys stay in registers/LDS.
// The plan is integer work of ~ (8 + 8*passes) bytes per lookup per core; its cost is launch
// latency and dependent-load chains, not bandwidth -- hence the effort to keep it to one launch.
#include <stdlib.h>

#include <type_traits>

#include "ttx_internal.h"

namespace ttx {

constexpr int kPlanThreads = 1024;
constexpr int kPlanWaves = kPlanThreads / kWave;

int max_chunks(const Dims& d, long long nnz, int MC) {
  if (MC <= 0) MC = 1;  // (a shape no kernel variant fits: the sizes stay defined, the launch reports TTX_EUNSUPPORTED)
  long long a = nnz < d.S[1] ? nnz : d.S[1];
  long long v = a + nnz / MC + 1;
  return (int)v;
}

static size_t r64(size_t k) { return (k + 63) / 64 * 64; }

constexpr int kMaxGroupsHost = 32;  // == kMaxGroups (table groups of the wide plan, below)
// digit counts of the multi-work-group plans: a 256-entry row per wave unit of 256 positions, or
// (wide-digit plan) a 2048-entry row per work-group of 4096 positions
static size_t cnt_ints(const Dims& d, long long nnz) {  // (+ a row per table group of the grouped wide plan)
  return (size_t)d.T * (256 * ((nnz + 255) / 256 + 1) + 2048 * (kMaxGroupsHost + 2));
}

static size_t plan_ints(const Dims& d, long long nnz, int MC) {
  size_t n = r64(64);                                               // hdr
  for (int t = 0; t < d.T; ++t) n += r64(nnz) * 6 + r64((size_t)d.S[t] + 1);  // sid, perm, ipos, 3 scratch, off
  n += r64((size_t)d.S[1] + 1);                                     // chunk_off
  n += r64((size_t)max_chunks(d, nnz, MC) * 4);                     // chunk_rec
  n += r64((size_t)nnz * 4);                                        // lrec
  n += r64((size_t)nnz);                                            // lrow
  n += r64(cnt_ints(d, nnz));                                       // multi-block digit counts
  if (const size_t tf = t4_scratch_floats(d)) n += 2 * r64((size_t)nnz * tf) + r64((size_t)nnz * 4);  // four cores on the three-core kernels: M, d M, the lookups in core 2's order
  return n;
}

size_t plan_bytes(const Dims& d, long long nnz) {
  return align_up(plan_ints(d, nnz, choose_chunk(d, nnz)) * sizeof(int));
}

Plan carve_plan(const Dims& d, long long nnz, void* base) {
  Plan P;
  memset(&P, 0, sizeof(P));
  P.MC = choose_chunk(d, nnz);
  P.max_chunks = max_chunks(d, nnz, P.MC);
  int* cur = (int*)base;
  auto take = [&](size_t k) { int* r = cur; cur += r64(k); return r; };
  P.hdr = take(64);
  P.chunk_rec = (int4*)take((size_t)P.max_chunks * 4);
  P.lrec = (int4*)take((size_t)nnz * 4);
  P.lrow = take(nnz);
  P.cnt = take(cnt_ints(d, nnz));
  for (int t = 0; t < d.T; ++t) {
    P.sid[t] = take(nnz);
    P.perm[t] = take(nnz);
    P.ipos[t] = take(nnz);
    P.off[t] = take((size_t)d.S[t] + 1);
    for (int i = 0; i < 3; ++i) P.scratch[t][i] = take(nnz);
  }
  P.chunk_off = take((size_t)d.S[1] + 1);
  if (const size_t tf = t4_scratch_floats(d)) {
    P.t4m = (float*)take((size_t)nnz * tf);
    P.t4g = (float*)take((size_t)nnz * tf);
    P.t4o = (int4*)take((size_t)nnz * 4);
  }
  return P;
}

// block-wide exclusive scan of one int per thread (1024 threads); returns the
// exclusive prefix, *total = block sum.  wtot: LDS int[kPlanWaves + 1].
__device__ __forceinline__ int block_excl_scan(int v, int* wtot, int* total) {
  const int lane = lane_id();
  const int w = threadIdx.x / kWave;
  int inc = wave_incl_scan(v);
  if (lane == kWave - 1) wtot[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < kPlanWaves; ++i) { int c = wtot[i]; wtot[i] = run; run += c; }
    wtot[kPlanWaves] = run;
  }
  __syncthreads();
  int res = wtot[w] + inc - v;
  *total = wtot[kPlanWaves];
  __syncthreads();
  return res;
}

// i_t of lookup idx (reference decode: i_t = idx / L_t % p_t, cu:795-799; out-of-range
// indices, which the reference reads out of bounds with, are clamped).  When
// prod(p) <= 2^32 the int64 divisions become two 32-bit magic-number divisions.
struct CoreDec {  // everything the decode of one core needs, fetched from the kernarg once
  UDiv dl, dp;
  long long L;
  int p, first, idx32;
};
__device__ __forceinline__ CoreDec core_dec(const Dims& d, int t) {
  CoreDec c;
  c.dl = d.dvL[t];
  c.dp = d.dvP[t];
  c.L = d.L[t];
  c.p = d.p[t];
  c.first = (t == 0);
  c.idx32 = d.idx32;
  return c;
}
__device__ __forceinline__ int decode_core(const CoreDec& c, long long idx) {
  if (idx < 0) idx = 0;
  if (c.idx32 && (unsigned long long)idx < (1ull << 32)) {
    const unsigned q = udiv((unsigned)idx, c.dl);
    const unsigned a = c.first ? q : q - udiv(q, c.dp) * c.dp.d;
    return (int)min(a, (unsigned)(c.p - 1));
  }
  long long a = idx / c.L;
  if (!c.first) a = a % c.p;
  if (a >= c.p) a = c.p - 1;
  return (int)a;
}
__device__ __forceinline__ int decode_core(const Dims& d, int t, long long idx) {
  return decode_core(core_dec(d, t), idx);
}
// slice id of (table tb, index idx) in core t: tb * p_t + i_t -- or, tables of different row
// factors (Dims::tab), base_t[tb] + i_t with the table's own factors (plain divisions: only the
// wide-digit and the multi-pass plan take such a geometry)
__device__ __forceinline__ int slice_id(const Dims& d, const CoreDec& ct, int t, int tb, long long idx) {
  if (!d.tab) return tb * ct.p + decode_core(ct, idx);
  tb = min(max(tb, 0), d.num_tables - 1);
  const int p = d.tab->p[tb][t];
  const long long L = d.tab->L[tb][t];
  if (idx < 0) idx = 0;
  long long a;
  if ((((unsigned long long)idx | (unsigned long long)L) >> 32) == 0) a = (unsigned)idx / (unsigned)L;
  else a = idx / L;
  if (t > 0) a %= p;
  if (a >= p) a = p - 1;
  return d.tab->base[tb][t] + (int)a;
}
// branch-free 32-bit decode (prod(p) <= 2^32; idx already clamped to [0, 2^32))
__device__ __forceinline__ int decode32(const CoreDec& c, unsigned idx) {
  const unsigned q = udiv(idx, c.dl);
  const unsigned a = c.first ? q : q - udiv(q, c.dp) * c.dp.d;
  return (int)min(a, (unsigned)(c.p - 1));
}
__device__ __forceinline__ unsigned clamp_idx32(long long idx) {
  return idx < 0 ? 0u : (idx > 0xffffffffll ? 0xffffffffu : (unsigned)idx);
}

// ---- tiny-batch path (nnz <= 1024; the template extents still allow 16384): everything between the index load and
// the final stores stays on chip.  Each thread keeps its (key, value, rank) triples in
// registers (wave w owns the contiguous range [w*per, (w+1)*per) of the current
// order, 64 per batch), passes exchange through LDS, and the pivot work-group emits the
// flat lookup records at scatter time.  One global round trip in, one out.

template <int kBPW>  // batches of 64 lookups per wave (register array extent)
__global__ __launch_bounds__(kPlanThreads) void plan_small_kernel(
    Dims d, int N, const int64_t* __restrict__ indices, const int64_t* __restrict__ tableidx,
    const int64_t* __restrict__ rowidx, Plan P, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) int lds[];
#define PSTAMP(i) do { if (stamps && threadIdx.x == 0) stamps[(size_t)(1000 + blockIdx.x) * 16 + (i)] = wall_clock64(); } while (0)
  PSTAMP(0);
  int* hist = lds;                        // [256][kPlanWaves]
  int* wtot = hist + 256 * kPlanWaves;    // [kPlanWaves + 1] (+pad)
  int* keyL = wtot + 32;                  // [N]
  int* valL = keyL + ((N + 63) / 64 * 64);  // [N]
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int w = tid / kWave;
  const int t = blockIdx.x;
  const int per = ((N + kPlanWaves - 1) / kPlanWaves + kWave - 1) / kWave * kWave;
  const int nb = per / kWave;  // <= kBPW
  const int bits = 32 - __clz(max(d.S[t] - 1, 1));
  const int passes = max((bits + 7) / 8, 1);
  // single-pass pivot sort: the flat records go out straight from registers at scatter time
  const bool direct = (t == 1) && (passes == 1);
  const int wbeg = w * per;
  const int wend = min(N, wbeg + per);

  int k[kBPW], v[kBPW], r[kBPW];
  // direct path: the other cores' slice ids wait in global scratch (written and read
  // back by the same thread) so the register arrays stay within the 128-VGPR budget
  int* sc0 = P.scratch[1][0];
  int* sc2 = P.scratch[1][1];
  int* sc3 = P.scratch[1][2];
  // ---- 1. decode (coalesced).  All loads are issued before any dependent work:
  // out-of-range lanes re-read element N-1 instead of branching around the load.
  {
    unsigned idx[kBPW];
    int tbv[kBPW];
    const int last_i = N > 0 ? N - 1 : 0;
    const bool have_tb = tableidx != nullptr;
#pragma unroll
    for (int b = 0; b < kBPW; ++b) {
      const int i = min(wbeg + b * kWave + lane, last_i);
      idx[b] = (N > 0) ? clamp_idx32(indices[i]) : 0u;
      tbv[b] = (N > 0 && have_tb) ? (int)tableidx[i] : 0;
    }
    __builtin_amdgcn_sched_barrier(0);  // keep every load ahead of the arithmetic
    const CoreDec ct = core_dec(d, t), c0 = core_dec(d, 0), c2 = core_dec(d, 2), c3 = core_dec(d, 3);
#pragma unroll
    for (int b = 0; b < kBPW; ++b) {
      const int i = wbeg + b * kWave + lane;
      v[b] = i;
      k[b] = tbv[b] * ct.p + decode32(ct, idx[b]);
      if (direct && b < nb && i < wend) {
        sc0[i] = tbv[b] * c0.p + decode32(c0, idx[b]);
        if (d.T > 2) sc2[i] = tbv[b] * c2.p + decode32(c2, idx[b]);
        if (d.T > 3) sc3[i] = tbv[b] * c3.p + decode32(c3, idx[b]);
      }
    }
  }
  PSTAMP(1);
  // ---- 2. stable LSD radix sort, 8 bits per pass ----------------------------
  for (int ps = 0; ps < passes; ++ps) {
    const int shift = ps * 8;
    for (int e = tid; e < 256 * kPlanWaves; e += kPlanThreads) hist[e] = 0;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < kBPW; ++b) {
      if (b < nb) {  // wave-uniform
        const bool valid = wbeg + b * kWave + lane < wend;
        const unsigned dg = ((unsigned)k[b] >> shift) & 255u;
        const unsigned long long peers = wave_match8(dg, valid);
        if (valid) {
          const int before = hist[dg * kPlanWaves + w];
          r[b] = before + __popcll(peers & lanemask_lt());
          if ((peers & lanemask_lt()) == 0) hist[dg * kPlanWaves + w] = before + __popcll(peers);
        }
      }
    }
    PSTAMP(6);
    __syncthreads();
    {
      int v0 = hist[tid * 4 + 0], v1 = hist[tid * 4 + 1], v2 = hist[tid * 4 + 2], v3 = hist[tid * 4 + 3];
      int total;
      int ex = block_excl_scan(v0 + v1 + v2 + v3, wtot, &total);
      hist[tid * 4 + 0] = ex;
      hist[tid * 4 + 1] = ex + v0;
      hist[tid * 4 + 2] = ex + v0 + v1;
      hist[tid * 4 + 3] = ex + v0 + v1 + v2;
    }
    __syncthreads();
    PSTAMP(7);
    const bool last = (ps == passes - 1);
    if (direct) {
      // single pass: v[b] is still this thread's own element; read its other slice ids
      // back (all loads first), then scatter keys, values and the flat records
      int a0[kBPW], a2[kBPW], a3[kBPW], rw[kBPW];
      const int last_i = N > 0 ? N - 1 : 0;
#pragma unroll
      for (int b = 0; b < kBPW; ++b) {
        const int i = min(v[b], last_i);
        rw[b] = (b < nb && rowidx) ? (int)rowidx[i] : 0;
        a0[b] = (b < nb) ? sc0[i] : 0;
        a2[b] = (b < nb && d.T > 2) ? sc2[i] : 0;
        a3[b] = (b < nb && d.T > 3) ? sc3[i] : 0;
      }
#pragma unroll
      for (int b = 0; b < kBPW; ++b) {
        if (b < nb && wbeg + b * kWave + lane < wend) {
          const unsigned dg = ((unsigned)k[b] >> shift) & 255u;
          const int pos = hist[dg * kPlanWaves + w] + r[b];
          keyL[pos] = k[b];
          valL[pos] = v[b];
          P.lrec[pos] = make_int4(v[b], a0[b], a2[b], a3[b]);
          if (rowidx) P.lrow[pos] = rw[b];
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < kBPW; ++b) {
        if (b < nb && wbeg + b * kWave + lane < wend) {
          const unsigned dg = ((unsigned)k[b] >> shift) & 255u;
          const int pos = hist[dg * kPlanWaves + w] + r[b];
          keyL[pos] = k[b];
          valL[pos] = v[b];
        }
      }
    }
    __syncthreads();
    if (!last) {
#pragma unroll
      for (int b = 0; b < kBPW; ++b) {
        const int i = wbeg + b * kWave + lane;
        if (b < nb && i < wend) { k[b] = keyL[i]; v[b] = valL[i]; }
      }
      __syncthreads();
    }
  }
  PSTAMP(2);
  // ---- 3. stores: perm, slice offsets (run heads of the sorted keys) -----------
  if (t != 1) {  // the pivot's consumers read lrec / chunk_rec / chunk_off only
    int* pm = P.perm[t];
    for (int i = tid; i < N; i += kPlanThreads) { pm[i] = valL[i]; P.ipos[t][valL[i]] = i; }
    int* off = P.off[t];
    const int S = d.S[t];
    for (int i = tid; i <= N; i += kPlanThreads) {
      const int kprev = (i == 0) ? -1 : keyL[i - 1];
      const int kcur = (i == N) ? S : keyL[i];
      for (int s = kprev + 1; s <= kcur; ++s) off[s] = i;
    }
  }
  PSTAMP(3);
  if (t != 1) return;
  // ---- 4. pivot core: flat per-lookup records + chunk work-list ----------------
  {
    const CoreDec c0 = core_dec(d, 0), c2 = core_dec(d, 2), c3 = core_dec(d, 3);
    for (int i = tid; i < N && !direct; i += kPlanThreads) {
      const int n = valL[i];
      const unsigned idx = clamp_idx32(indices[n]);
      const int tb = tableidx ? (int)tableidx[n] : 0;
      int4 rr;
      rr.x = n;
      rr.y = tb * c0.p + decode32(c0, idx);
      rr.z = d.T > 2 ? tb * c2.p + decode32(c2, idx) : 0;
      rr.w = d.T > 3 ? tb * c3.p + decode32(c3, idx) : 0;
      P.lrec[i] = rr;
      if (rowidx) P.lrow[i] = (int)rowidx[n];
    }
    PSTAMP(4);
    // chunk list from the sorted keys in LDS: slice s covers [lower_bound(s), lower_bound(s+1))
    const int S1 = d.S[1];
    const int MC = P.MC;
    int carry = 0;
    for (int s0 = 0; s0 < S1; s0 += kPlanThreads) {
      const int s = s0 + tid;
      int nch = 0, beg = 0, cnt = 0;
      if (s < S1) {
        int lo = 0, hi = N;  // first position with key >= s
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keyL[mid] < s) lo = mid + 1; else hi = mid; }
        beg = lo;
        hi = N;              // first position with key >= s + 1
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keyL[mid] <= s) lo = mid + 1; else hi = mid; }
        cnt = lo - beg;
        nch = (cnt + MC - 1) / MC;
      }
      int total;
      const int ex = carry + block_excl_scan(nch, wtot, &total);
      if (s < S1) {
        P.chunk_off[s] = ex;
        for (int j = 0; j < nch; ++j)
          P.chunk_rec[ex + j] = make_int4(s, beg + j * MC, min(MC, cnt - j * MC), ex + j);
      }
      carry += total;
    }
    for (int c = carry + tid; c < P.max_chunks; c += kPlanThreads) P.chunk_rec[c] = make_int4(0, 0, 0, 0);
    if (tid == 0) {
      P.chunk_off[S1] = carry;
      P.hdr[0] = carry;
      P.hdr[1] = MC;
      P.hdr[2] = N;
      P.hdr[3] = rowidx ? 1 : 0;
      P.hdr[kHdrT4Valid] = 0;  // (a new plan: no merged last cores yet)
      for (int tt = 0; tt < TTX_MAX_CORES; ++tt) P.hdr[8 + tt] = -1;  // hot slices per core: unknown (reduce_apply looks)
    }
    PSTAMP(5);
  }
#undef PSTAMP
}

// ---- single 8-bit pass on wave units (every S[t] <= 256, 16384 < nnz <= ~1 M) ----
// A "unit" is one wavefront walking A.unit consecutive positions in batches of 64 (ranking by
// wave_match8, running digit counters in LDS):
//   mb_count   : per-unit digit counts -> cnt[t][unit][digit]
//   mb_scatter : every work-group scans the unit counts itself (<= kMbFuseU units), position =
//                base[digit][unit] + rank; writes perm[t] / the pivot's flat lookup records; its
//                first work-group per core writes offsets and chunk list (finish_single_pass).
// (Several 8-bit passes: mbp_* below.)
constexpr int kMbThreads = 256;            // 4 wave units per work-group
constexpr int kMbUnits = kMbThreads / kWave;
constexpr int kSB = 4;                     // batches of 64 whose loads are in flight together
constexpr int kMbFuseU = 256;              // up to this many units the scatter pass scans the counts itself

struct MbArgs {
  int N, U;            // lookups (upper bound when n_dev is set), units per core
  const int* n_dev;    // device-side lookup count (<= N), or NULL: N is exact
  int unit;            // positions per wave unit (multiple of 64)
  int* cnt;            // [T][U][256]  (unit-major: a unit's 256 digit counts are one 1 KiB row)
};

// the number of lookups a kernel works on: the host value, or the device-side count clamped to it
__device__ __forceinline__ int live_n(int n_host, const int* n_dev) {
  if (!n_dev) return n_host;
  const int v = *n_dev;
  return v < 0 ? 0 : (v < n_host ? v : n_host);
}

// what one lane holds of one batch of 64 positions
struct MbItem { int val, kv; };

// pass 0 reads (and decodes) the indices; later passes chase order -> key
__device__ __forceinline__ MbItem mb_load(int i, bool valid, int pass, const Dims& d, int t, const CoreDec& ct,
                                          const int64_t* __restrict__ indices, const int64_t* __restrict__ tableidx,
                                          const int* __restrict__ src, const int* __restrict__ key) {
  MbItem it{0, 0};
  if (!valid) return it;
  if (pass == 0) {
    const int tb = tableidx ? (int)tableidx[i] : 0;
    it.val = i;
    it.kv = slice_id(d, ct, t, tb, indices[i]);
  } else {
    it.val = src[i];
    it.kv = key[it.val];
  }
  return it;
}

__global__ __launch_bounds__(kMbThreads) void mb_count_kernel(
    Dims d, MbArgs A, const int64_t* __restrict__ indices, const int64_t* __restrict__ tableidx) {
  __shared__ int hist[kMbUnits][256];
  const int t = blockIdx.y;
  const int lane = lane_id(), w = threadIdx.x / kWave;
  const int u = blockIdx.x * kMbUnits + w;
  for (int e = lane; e < 256; e += kWave) hist[w][e] = 0;
  const int N = live_n(A.N, A.n_dev);
  const int beg = min(N, u * A.unit), end = min(N, beg + A.unit);
  const CoreDec ct = core_dec(d, t);
  // super-batches of kSB x 64 positions: all their (dependent) loads are issued before any is used --
  // a wave walks thousands of positions and one round trip per 64 was the whole cost of the pass
  for (int base = beg; base < end; base += kSB * kWave) {
    MbItem it[kSB];
#pragma unroll
    for (int k = 0; k < kSB; ++k) {
      const int i = base + k * kWave + lane;
      it[k] = mb_load(i, i < end, 0, d, t, ct, indices, tableidx, nullptr, nullptr);
    }
#pragma unroll
    for (int k = 0; k < kSB; ++k) {
      const int i = base + k * kWave + lane;
      const bool valid = i < end;
      const unsigned dg = (unsigned)it[k].kv & 255u;
      const unsigned long long peers = wave_match8(dg, valid);
      if (valid && (peers & lanemask_lt()) == 0) hist[w][dg] += __popcll(peers);
    }
  }
  if (u < A.U)
    for (int e = lane; e < 256; e += kWave) A.cnt[((size_t)t * A.U + u) * 256 + e] = hist[w][e];
}

template <typename T>
__device__ __forceinline__ T* shifted(T* p, long long bytes) { return (T*)((char*)p + bytes); }

// Single 8-bit pass (every S[t] <= 256): digit == slice id, so the digit prefix IS the slice
// offset table and the pivot's chunk list follows from the digit totals -- what mb_finish would
// recompute from the sorted keys.  One 256-thread work-group per core, thread = digit dg;
// tot = lookups of the slice, dbase = its first position.  wt5: LDS int[kMbUnits + 1].
__device__ __forceinline__ void finish_single_pass(const Dims& d, int t, int dg, int tot, int dbase, int N,
                                                   bool has_row, const Plan& P, int* wt5, long long psh = 0) {
  int* const hdr = (int*)((char*)P.hdr + psh);
  int* const chunk_off = (int*)((char*)P.chunk_off + psh);
  int4* const chunk_rec = (int4*)((char*)P.chunk_rec + psh);
  const int lane = lane_id(), w = threadIdx.x / kWave;
  const int S = d.S[t];
  if (t != 1) {
    const int nhot = __syncthreads_count(dg < S && tot > 2 * kSegThin);  // (reduce_apply's hot slices, ttx_internal.h)
    if (dg == 0) hdr[8 + t] = nhot;
    if (dg <= S) shifted(P.off[t], psh)[dg] = (dg == S) ? N : dbase;
    if (dg == 0 && S == 256) shifted(P.off[t], psh)[256] = N;
    return;
  }
  // chunk SLOTS (where a chunk's d core_1 partial lives) are contiguous per slice; the DISPATCH
  // order (index into chunk_rec = blockIdx of the contraction kernels) puts all full chunks
  // first and the partial ones after: work-groups go round-robin over the 8 XCDs, and a
  // slice-major list (full, partial, full, partial, ..) would hand every full chunk to the
  // even XCDs.  Largest-first is also the order a greedy dispatcher balances best.
  const int MC = P.MC;
  const int nf = dg < S ? tot / MC : 0;            // full chunks
  const int pr = dg < S ? tot - nf * MC : 0;       // lookups in the partial chunk
  const int np = pr ? 1 : 0;
  const int nhot = __syncthreads_count(nf + np > kHotRowsPivot);
  if (dg == 0) hdr[8 + 1] = nhot;
  const int packed = (nf << 12) | np;              // slices <= 256: np sums stay < 4096
  const int cinc = wave_incl_scan(packed);
  __syncthreads();
  if (lane == kWave - 1 && w < kMbUnits) wt5[w] = cinc;  // (digits live in the first 256 threads)
  __syncthreads();
  int cb = 0, call = 0;
  for (int k = 0; k < kMbUnits; ++k) { const int v = wt5[k]; if (k < w) cb += v; call += v; }
  const int exq = cb + cinc - packed;
  const int fbase = exq >> 12, pbase = exq & 4095;  // full / partial chunks of earlier slices
  const int ftot = call >> 12, ptot = call & 4095;
  const int ctot = ftot + ptot;
  const int ex = fbase + pbase;                     // first slot of this slice
  // the records are written by ALL threads, one full chunk each per round (a skewed stream puts hundreds of full
  // chunks on one slice: its thread writing them one after the other was the tail of the launch): slice dg
  // publishes {first full chunk, first position, first slot} and every thread finds its chunk's slice by binary
  // search over the full-chunk prefix
  __shared__ int cf_base[257], cf_pos[256], cf_slot[256];
  if (dg < S) {
    chunk_off[dg] = ex;
    cf_base[dg] = fbase;
    cf_pos[dg] = dbase;
    cf_slot[dg] = ex;
    if (np) chunk_rec[ftot + pbase] = make_int4(dg, dbase + nf * MC, pr, ex + nf);
  }
  if (dg == 0) cf_base[S] = ftot;
  __syncthreads();
  for (int c = threadIdx.x; c < ftot; c += blockDim.x) {
    int lo = 0, hi = S;  // the last slice with cf_base <= c (slices without full chunks share their successor's base)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cf_base[mid] <= c) lo = mid; else hi = mid;
    }
    const int j = c - cf_base[lo];
    chunk_rec[c] = make_int4(lo, cf_pos[lo] + j * MC, MC, cf_slot[lo] + j);
  }
  for (int cc = ctot + dg; cc < P.max_chunks; cc += blockDim.x) chunk_rec[cc] = make_int4(0, 0, 0, 0);
  if (dg == 0) {
    chunk_off[S] = ctot;
    hdr[0] = ctot;
    hdr[1] = MC;
    hdr[2] = N;
    hdr[3] = has_row ? 1 : 0;
    hdr[kHdrT4Valid] = 0;
  }
}

// Small batches whose sorts are all single-pass: ONE launch builds the whole plan.  Every
// work-group histograms all N keys of its core itself (LDS integer atomics -- N/256 decodes per
// thread is cheaper than a count launch), then ranks and scatters its own 4 x kOneUnit positions.
constexpr int kOneUnit = 64;      // positions per wave
constexpr int kOneMaxN = 16384;
// PRO: the launch is also the lookup PROLOGUE of a one-table module (tableidx == 0): bag rows
// from the offsets (compute_rowidx_kernel, cu:1338-1354, as a binary search over an LDS copy of
// the offsets) and the hash-table frequency update (update_cache_state, cu:1077-1113), both done
// by the work-groups of core 0 for their own positions.
constexpr int kProMaxBags = 4096;
struct Prologue {
  const int64_t* offsets;  // [nb + 1]
  int nb;
  int64_t* rowidx;         // out [N]
  int64_t* tableidx;       // out [N] (zeros)
  int H;                   // 0: no frequency update
  int64_t* hashtbl;
  int64_t* cache_freq;
};
#ifndef TTX_PLAN_XWG
#define TTX_PLAN_XWG 1
#endif
constexpr int kOneThreads = 1024;                 // 16 waves: the histogram of all N keys is 4x shorter per thread
constexpr int kOneWaves = kOneThreads / kWave;
template <bool PRO>
__global__ __launch_bounds__(kOneThreads) void mb_single_kernel(
    Dims d, int Nmax, const int* __restrict__ n_dev, const int64_t* __restrict__ indices,
    const int64_t* __restrict__ tableidx, const int64_t* __restrict__ rowidx, Plan P, Prologue pg, ProBatch mb) {
  if (PRO && blockIdx.z > 0) {  // (work-group-uniform) a later batch of a multi-batch launch
    const int z = blockIdx.z;
    TTX_PICK_BATCH(mb, z, indices, pg.offsets)
    pg.rowidx += (long long)z * mb.out_stride;
    pg.tableidx += (long long)z * mb.out_stride;
  }
  if (!PRO && blockIdx.z > 0) {  // batch z of plan_build_batches: its arrays lie at a constant stride behind batch 0's
    const long long sh = (long long)blockIdx.z * mb.out_stride;
    indices += sh;
    if (tableidx) tableidx += sh;
    if (rowidx) rowidx += sh;
    if (n_dev) n_dev += blockIdx.z;
  }
  // batch z's plan lies z * plan_stride bytes behind batch 0's: applied where a pointer is used (modifying the by-value
  // Plan would make it a local copy, and its run-time subscripts P.perm[t] .. would go through scratch memory)
  const long long psh = (long long)blockIdx.z * mb.plan_stride;
  // (a ballot-grouped histogram -- wave_match8 per batch, the group's first lane adding the group size to a
  //  wave-private row -- was measured against these LDS atomics: 17.5 vs 12.5 us uniform, 30.7 vs 28.3 us on a
  //  skewed stream: the ~60 VALU instructions of a match cost more than the atomics' conflicts)
  // one LDS atomic per key: digit counts of the keys in front of this work-group's window (hbef), inside it per wave
  // slot (hrun) and behind it (haft); a digit's total is their sum
  __shared__ int haft[256], hbef[256], hrun[kOneWaves][256];
  const int N = live_n(Nmax, n_dev);
  __shared__ int wt5[kMbUnits + 1];
  __shared__ int offs[PRO ? kProMaxBags + 1 : 1];
  const int t = blockIdx.y, tid = threadIdx.x;
  const int lane = lane_id(), w = tid / kWave;
  const CoreDec ct = core_dec(d, t);
  if (PRO) tableidx = nullptr;
  // ---- every global load of the launch is issued here: all N keys (batch m = positions [1024 m, 1024 m + 1024),
  // this thread's key of the batch at 1024 m + tid), then the offsets
  constexpr int kMaxB = kOneMaxN / kOneThreads;  // 16 batches
  const int nbat = (N + kOneThreads - 1) / kOneThreads;
  long long ix[kMaxB];
  int tb[kMaxB];
#pragma unroll
  for (int m = 0; m < kMaxB; ++m) {
    const int i = m * kOneThreads + tid;
    ix[m] = (m < nbat && i < N) ? indices[i] : 0;
    tb[m] = (m < nbat && i < N && tableidx) ? (int)tableidx[i] : 0;
  }
  if (PRO) {
    if (t <= 1)  // only core 0 (rowidx out) and the pivot (lrow) need bag rows
      for (int e = tid; e <= pg.nb; e += kOneThreads) offs[e] = (int)min(pg.offsets[e], (int64_t)0x7fffffff);
  }
  if (tid < 256) { haft[tid] = 0; hbef[tid] = 0; }
  for (int e = tid; e < kOneWaves * 256; e += kOneThreads) (&hrun[0][0])[e] = 0;
  __syncthreads();
  const int bx = blockIdx.x;
  const int bbeg = bx * (kOneWaves * kOneUnit), bend = min(N, bbeg + kOneWaves * kOneUnit);
  // this thread's own position (the one it ranks and scatters): batch bx
  const int i = bbeg + tid;
  const bool valid = i < bend;
  int kv = 0, tbv = 0, row = 0;
  long long idx = 0;
  bool peel_on = true;
#pragma unroll
  for (int m = 0; m < kMaxB; ++m) {
    if (m < nbat) {  // (work-group-uniform)
      const int im = m * kOneThreads + tid;
      const bool vm = im < N;
      const int km = vm ? min(tb[m] * ct.p + decode_core(ct, ix[m]), 255) : 0;  // tableidx is not validated
      // Same-address LDS atomics of one wave instruction serialise, and a skewed stream puts most of a batch on one
      // digit (cfg3's: 90 % of core 0's keys, 70 % of the pivot's): the wave then peels the first lane's digit off --
      // one ballot, that lane adds the whole group's count -- and leaves the rest to one atomic per lane (plan kernel
      // 21.3 -> 14.2 us on cfg3's stream).  The ballot costs 1.4 us per launch on a uniform stream, so a wave peels
      // only while its previous batch had a group of >= 12 lanes (its first batch always tries).  A full ballot
      // grouping of all digits (wave_match8) costs more than it saves.
      unsigned long long rest = __ballot(vm);
      if (peel_on && rest != 0) {  // (wave-uniform)
        const int src = __ffsll((long long)rest) - 1;
        const int d0 = __shfl(km, src, kWave);
        const unsigned long long grp = __ballot(vm && km == d0) & rest;
        const int c = __popcll(grp);
        if (lane == src) atomicAdd(m < bx ? &hbef[d0] : (m == bx ? &hrun[w][d0] : &haft[d0]), c);
        rest &= ~grp;
        peel_on = c >= 12;
      }
      if ((rest >> lane) & 1ull) atomicAdd(m < bx ? &hbef[km] : (m == bx ? &hrun[w][km] : &haft[km]), 1);
      if (vm && m == bx) { kv = km; idx = ix[m]; tbv = tb[m]; }
    }
  }
  // Frequency update of this wave's 64 keys (work-groups of the last core, which have no bag rows to find),
  // split in two so that its CAS round trip (random 8-byte slots in HBM) overlaps the work below:
  // here equal keys of the wave are combined (hashtbl_count_wave's grouping) and the group leaders issue
  // the CAS; its result is looked at before the scatter.
  bool h_lead = false;
  long long h_key = 0;
  unsigned long long h_times = 0, h_old = 0;
  int h_idx = 0;
  const bool h_on = PRO && pg.H && t == (d.T >= 3 ? 2 : 0);
  if (h_on) {
    h_key = valid ? idx : 0;
    const unsigned h = valid ? hash64(h_key, pg.H) : 0u;
    const unsigned long long peers = wave_match8(h & 255u, valid);
    const int leader = valid ? __ffsll((long long)peers) - 1 : 0;
    const int klo = __shfl((int)(unsigned)h_key, leader, kWave), khi = __shfl((int)(h_key >> 32), leader, kWave);
    const bool eq = valid && klo == (int)(unsigned)h_key && khi == (int)(h_key >> 32);
    const unsigned long long eqm = __ballot(eq);
    if (valid && (!eq || lane == leader)) {
      h_lead = true;
      h_times = eq ? (unsigned long long)__popcll(peers & eqm) : 1ull;
      h_idx = (int)h;
      h_old = atomicCAS((unsigned long long*)&pg.hashtbl[h_idx], (unsigned long long)(-1ll), (unsigned long long)h_key);
    }
  }
  if (valid && PRO && t <= 1) {  // bag of position i: the last b with offsets[b] <= i (empty bags skipped)
    int lo = 0, hi = pg.nb;  // answer in [lo, hi)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offs[mid] <= i) lo = mid; else hi = mid;
    }
    row = lo;
    if (t == 0) {
      pg.rowidx[i] = row;
      pg.tableidx[i] = 0;
    }
  }
  __syncthreads();
  // digit dg = tid (first 256 threads): total, exclusive prefix over digits, first position per wave
  const int dg = tid;
  const bool isd = tid < 256;
  int mine[kOneWaves];
#pragma unroll
  for (int k = 0; k < kOneWaves; ++k) mine[k] = isd ? hrun[k][dg] : 0;
  const int bef = isd ? hbef[dg] : 0;
  int tot = isd ? bef + haft[dg] : 0;
#pragma unroll
  for (int k = 0; k < kOneWaves; ++k) tot += mine[k];
  const int inc = wave_incl_scan(tot);
  if (isd && lane == kWave - 1) wt5[w] = inc;
  __syncthreads();
  int wbase = 0;
  if (isd)
    for (int k = 0; k < w; ++k) wbase += wt5[k];
  const int dbase = wbase + inc - tot;
  if (isd) {
    int b = dbase + bef;
#pragma unroll
    for (int k = 0; k < kOneWaves; ++k) { hrun[k][dg] = b; b += mine[k]; }
  }
  __syncthreads();
  // offsets / chunk list: by an EXTRA work-group per core (the last of the grid, whose window lies behind the batch), so
  // that no work-group has both the list and 1024 positions to scatter on its critical path
  if (blockIdx.x == gridDim.x - 1) finish_single_pass(d, t, dg, tot, dbase, N, PRO || rowidx != nullptr, P, wt5, psh);
  // rank + scatter this wave's 64 positions
  if (h_lead) {  // second half of the frequency update: count, or keep probing (hashtbl_cuda_utils.cuh:102-133)
    for (int pr = 0;; ++pr) {
      if ((long long)h_old == -1 || (long long)h_old == h_key) {
        atomicAdd((unsigned long long*)&pg.cache_freq[h_idx], h_times);
        break;
      }
      if (pr == kMaxProbes - 1) break;  // dropped
      h_idx = (h_idx + 1) % pg.H;
      h_old = atomicCAS((unsigned long long*)&pg.hashtbl[h_idx], (unsigned long long)(-1ll), (unsigned long long)h_key);
    }
  }
  const unsigned long long peers = wave_match8((unsigned)kv, valid);
  if (valid) {
    const int pos = hrun[w][kv] + __popcll(peers & lanemask_lt());
    if (t != 1) {
      shifted(P.perm[t], psh)[pos] = i;
      shifted(P.ipos[t], psh)[i] = pos;
    } else {
      const int s0 = tbv * d.p[0] + decode_core(d, 0, idx);
      const int s2 = d.T > 2 ? tbv * d.p[2] + decode_core(d, 2, idx) : 0;
      const int s3 = d.T > 3 ? tbv * d.p[3] + decode_core(d, 3, idx) : 0;
      shifted(P.lrec, psh)[pos] = make_int4(i, s0, s2, s3);
      if (PRO) shifted(P.lrow, psh)[pos] = row;
      else if (rowidx) shifted(P.lrow, psh)[pos] = (int)rowidx[i];
    }
  }
}

__global__ __launch_bounds__(kMbThreads) void mb_scatter_kernel(
    Dims d, MbArgs A, const int64_t* __restrict__ indices, const int64_t* __restrict__ tableidx,
    const int64_t* __restrict__ rowidx, Plan P) {
  __shared__ int run[kMbUnits][256];
  __shared__ int wt5[kMbUnits + 1];
  const int t = blockIdx.y;
  const int lane = lane_id(), w = threadIdx.x / kWave;
  const int u = blockIdx.x * kMbUnits + w;
  {
    // every work-group derives its own bases from the raw counts (<= kMbFuseU units, no scan launch).
    // thread = digit: total of the digit, exclusive prefix over digits, prefix over earlier units
    const int dg = threadIdx.x;
    const int* c = A.cnt + (size_t)t * A.U * 256 + dg;
    const int u0 = blockIdx.x * kMbUnits;
    int tot = 0, before = 0, mine[kMbUnits];
#pragma unroll
    for (int k = 0; k < kMbUnits; ++k) mine[k] = 0;
    for (int ub = 0; ub < A.U; ub += 8) {
      int v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (ub + j < A.U) ? c[(size_t)(ub + j) * 256] : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int uu = ub + j;
        if (uu < u0) before += v[j];
#pragma unroll
        for (int k = 0; k < kMbUnits; ++k) if (uu == u0 + k) mine[k] = v[j];
        tot += v[j];
      }
    }
    const int inc = wave_incl_scan(tot);
    if (lane == kWave - 1) wt5[w] = inc;
    __syncthreads();
    int wbase = 0;
    for (int k = 0; k < w; ++k) wbase += wt5[k];
    const int dbase = wbase + inc - tot;  // first position of digit dg
    int b = dbase + before;               // first position of (digit dg, unit u0)
#pragma unroll
    for (int k = 0; k < kMbUnits; ++k) { run[k][dg] = b; b += mine[k]; }
    __syncthreads();
    if (blockIdx.x == 0) finish_single_pass(d, t, dg, tot, dbase, live_n(A.N, A.n_dev), rowidx != nullptr, P, wt5);
  }
  const int N = live_n(A.N, A.n_dev);
  const int beg = min(N, u * A.unit), end = min(N, beg + A.unit);
  const bool pivot = (t == 1);
  const CoreDec ct = core_dec(d, t);
  // the key is re-derived from the index (cheaper than a key array written by the count launch); the pivot
  // decodes the other cores' slice ids of its records from the same index
  for (int base = beg; base < end; base += kSB * kWave) {
    long long ix[kSB];
    int tb[kSB], brow[kSB];
#pragma unroll
    for (int k = 0; k < kSB; ++k) {
      const int i = base + k * kWave + lane;
      ix[k] = i < end ? indices[i] : 0;
      tb[k] = (i < end && tableidx) ? (int)tableidx[i] : 0;
      brow[k] = (pivot && rowidx && i < end) ? (int)rowidx[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < kSB; ++k) {
      const int i = base + k * kWave + lane;
      const bool valid = i < end;
      const unsigned dg = valid ? (unsigned)slice_id(d, ct, t, tb[k], ix[k]) & 255u : 0u;
      const unsigned long long peers = wave_match8(dg, valid);
      if (valid) {
        const int before = run[w][dg];
        const int pos = before + __popcll(peers & lanemask_lt());
        if (!pivot) {
          P.perm[t][pos] = i;
          P.ipos[t][i] = pos;
        } else {  // the pivot's final order lives in lrec.x
          const int s0 = slice_id(d, core_dec(d, 0), 0, tb[k], ix[k]);
          const int s2 = d.T > 2 ? slice_id(d, core_dec(d, 2), 2, tb[k], ix[k]) : 0;
          const int s3 = d.T > 3 ? slice_id(d, core_dec(d, 3), 3, tb[k], ix[k]) : 0;
          P.lrec[pos] = make_int4(i, s0, s2, s3);
          if (rowidx) P.lrow[pos] = brow[k];
        }
        if ((peers & lanemask_lt()) == 0) run[w][dg] = before + __popcll(peers);
      }
    }
  }
}

// ---- wide-digit plan: 256 < max S[t] <= 2048 (a few tables batched), up to kWideMaxG work-groups ----
// The slice id fits ONE digit of 10 / 11 bits, so the sort is a single pass and digit == slice as
// in mb_single_kernel: mbw_count leaves one digit histogram per work-group of 4096 positions,
// mbw_scatter sums the rows before its own (thread = digit), ranks its 16 x 256 positions
// against per-wave histograms it rebuilds in LDS, and its first work-group writes the slice
// offsets / chunk list.  Two launches and one pass over the indices instead of five and two.
constexpr int kWideThreads = 1024;
constexpr int kWideWaves = kWideThreads / kWave;
constexpr int kWideSpan = kWideThreads * kSB;   // positions per work-group (kSB x 64 per wave)
constexpr int kWideMaxG = 96;                   // every scatter work-group reads all count rows

// Table groups (more than 2048 slice ids in a core, bags table-major as the module's offsets make them):
// the tables are cut into groups of `gsz` consecutive tables whose slice ids fit one wide digit, every
// group is sorted by itself inside its own range of positions [offsets[k gsz B], offsets[(k+1) gsz B))
// (already contiguous: the table IS the high digit and the input is sorted by it), work-groups are
// dealt to the groups in order (each group at least one), and the pivot's chunk list is built
// afterwards from the offsets (mb_chunks_kernel).  Three launches and one pass over the indices
// for what the multi-pass plan does in seven launches and two passes.
constexpr int kMaxGroups = kMaxGroupsHost;
struct GrpArgs {
  const int64_t* offsets;  // [num_tables * B + 1], or NULL: one group, all positions
  int B, gsz, ngroups;
};
struct GrpMap {  // what a work-group learns about its group
  int k, tb0, pos0, beg, end, wg0, wg1;  // group, first table, first position, own span, the group's work-groups
};
// whole work-group; gpos / gwg: LDS int[kMaxGroups + 1].  Returns false for a work-group without a group.
__device__ __forceinline__ bool grp_map(const Dims& d, const GrpArgs& ga, int N, int* gpos, int* gwg, GrpMap* m) {
  const int tid = threadIdx.x;
  if (tid <= ga.ngroups) {
    const int tb = min(tid * ga.gsz, d.num_tables);
    gpos[tid] = (int)min(ga.offsets[(size_t)tb * ga.B], (int64_t)N);
  }
  __syncthreads();
  if (tid == 0) {
    int a = 0;
    for (int k = 0; k < ga.ngroups; ++k) {
      gwg[k] = a;
      const int len = gpos[k + 1] - gpos[k];
      a += len > 0 ? (len + kWideSpan - 1) / kWideSpan : 1;  // (an empty group still writes its slice offsets)
    }
    gwg[ga.ngroups] = a;
  }
  __syncthreads();
  const int g = blockIdx.x;
  if (g >= gwg[ga.ngroups]) return false;
  int k = 0;
  while (k + 1 < ga.ngroups && gwg[k + 1] <= g) ++k;
  m->k = k;
  m->tb0 = k * ga.gsz;
  m->pos0 = gpos[k];
  m->beg = min(gpos[k + 1], gpos[k] + (g - gwg[k]) * kWideSpan);
  m->end = min(gpos[k + 1], m->beg + kWideSpan);
  m->wg0 = gwg[k];
  m->wg1 = gwg[k + 1];
  return true;
}
// first slice id of table tb in core t
__device__ __forceinline__ int slice_base(const Dims& d, const CoreDec& ct, int t, int tb) {
  if (tb >= d.num_tables) return d.S[t];
  return d.tab ? d.tab->base[tb][t] : tb * ct.p;
}

template <int BITS, bool GRP>
__global__ __launch_bounds__(kWideThreads) void mbw_count_kernel(
    Dims d, int Nmax, const int* __restrict__ n_dev, const int64_t* __restrict__ indices,
    const int64_t* __restrict__ tableidx, int* __restrict__ cnt, GrpArgs ga) {
  constexpr int BINS = 1 << BITS;
  __shared__ int hist[BINS];
  __shared__ int gpos[GRP ? kMaxGroups + 1 : 1], gwg[GRP ? kMaxGroups + 1 : 1];
  const int t = blockIdx.y, tid = threadIdx.x;
  for (int e = tid; e < BINS; e += kWideThreads) hist[e] = 0;
  const int N = live_n(Nmax, n_dev);
  const CoreDec ct = core_dec(d, t);
  int beg = blockIdx.x * kWideSpan, end = min(N, beg + kWideSpan), sb = 0;
  if (GRP) {
    GrpMap m;
    if (!grp_map(d, ga, N, gpos, gwg, &m)) return;
    beg = m.beg;
    end = m.end;
    sb = slice_base(d, ct, t, m.tb0);
  } else {
    __syncthreads();
  }
  long long ix[kSB];
  int tb[kSB];
#pragma unroll
  for (int k = 0; k < kSB; ++k) {
    const int i = beg + k * kWideThreads + tid;
    ix[k] = i < end ? indices[i] : 0;
    tb[k] = (i < end && tableidx) ? (int)tableidx[i] : 0;
  }
#pragma unroll
  for (int k = 0; k < kSB; ++k) {
    const int i = beg + k * kWideThreads + tid;
    if (i < end) atomicAdd(&hist[min(max(slice_id(d, ct, t, tb[k], ix[k]) - sb, 0), BINS - 1)], 1);  // tableidx is not validated
  }
  __syncthreads();
  int* row = cnt + ((size_t)t * gridDim.x + blockIdx.x) * BINS;
  for (int e = tid; e < BINS; e += kWideThreads) row[e] = hist[e];
}

// offsets / chunk list from the digit totals (finish_single_pass for K digits per thread, 1024
// threads): thread owns digits tid * K + j.  wt: LDS int[kWideWaves].  Block 0 only, all threads.
template <int K>
__device__ __forceinline__ void finish_wide(const Dims& d, int t, const int (&tot)[K], const int (&dbase)[K], int N,
                                            bool has_row, const Plan& P, int* wt) {
  const int tid = threadIdx.x, lane = lane_id(), w = tid / kWave;
  const int S = d.S[t];
  if (t != 1) {
    // (round 6) the number of hot slices of this core, as finish_single_pass leaves it: with "unknown" (-1) every segment
    // work-group of reduce_apply copied the offset table and searched it to find that it has nothing to do -- the tail of the
    // launch on a uniform stream (two cores, 10k lookups: 15 of reduce_apply's 20 us; an integer sum: order-free)
    int nh = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int dg = tid * K + j;
      if (dg < S) P.off[t][dg] = dbase[j];
      nh += (dg < S && tot[j] > 2 * kSegThin) ? 1 : 0;
    }
    if (tid == 0) wt[0] = 0;
    __syncthreads();
    if (nh) atomicAdd(&wt[0], nh);
    __syncthreads();
    if (tid == 0) { P.off[t][S] = N; P.hdr[8 + t] = wt[0]; }
    return;
  }
  const int MC = P.MC;
  int nf[K], pr[K], packed[K], psum = 0;  // full chunks, lookups of the partial chunk, (full << 12) | partial
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int dg = tid * K + j;
    nf[j] = dg < S ? tot[j] / MC : 0;
    pr[j] = dg < S ? tot[j] - nf[j] * MC : 0;
    packed[j] = (nf[j] << 13) | (pr[j] ? 1 : 0);  // <= 4096 slices: the partial counts stay below 8192 (full chunks: < 2^18 on this route)
    psum += packed[j];
  }
  const int cinc = wave_incl_scan(psum);
  __syncthreads();
  if (lane == kWave - 1) wt[w] = cinc;
  __syncthreads();
  int cb = 0, call = 0;
  for (int k = 0; k < kWideWaves; ++k) { const int v = wt[k]; if (k < w) cb += v; call += v; }
  int exq = cb + cinc - psum;
  const int ftot = call >> 13, ptot = call & 8191;
  const int ctot = ftot + ptot;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int dg = tid * K + j;
    const int fbase = exq >> 13, pbase = exq & 8191;  // full / partial chunks of earlier slices
    const int ex = fbase + pbase;                     // first slot of this slice
    if (dg < S) {
      P.chunk_off[dg] = ex;
      for (int jj = 0; jj < nf[j]; ++jj) P.chunk_rec[fbase + jj] = make_int4(dg, dbase[j] + jj * MC, MC, ex + jj);
      if (pr[j]) P.chunk_rec[ftot + pbase] = make_int4(dg, dbase[j] + nf[j] * MC, pr[j], ex + nf[j]);
    }
    exq += packed[j];
  }
  for (int cc = ctot + tid; cc < P.max_chunks; cc += kWideThreads) P.chunk_rec[cc] = make_int4(0, 0, 0, 0);
  int nh = 0;  // hot pivot slices (reduce_apply: more chunk partials than kHotRowsPivot); the thin cores' blocks write their own word
#pragma unroll
  for (int j = 0; j < K; ++j) nh += (tid * K + j < S && nf[j] + (pr[j] ? 1 : 0) > kHotRowsPivot) ? 1 : 0;
  __syncthreads();  // (every thread is done with wt[])
  if (tid == 0) wt[0] = 0;
  __syncthreads();
  if (nh) atomicAdd(&wt[0], nh);
  __syncthreads();
  if (tid == 0) {
    P.chunk_off[S] = ctot;
    P.hdr[0] = ctot;
    P.hdr[1] = MC;
    P.hdr[2] = N;
    P.hdr[3] = has_row ? 1 : 0;
    P.hdr[kHdrT4Valid] = 0;
    // (the pivot: "none" or "some" -- with the COUNT known reduce_apply leaves more than kMaxHotPivot hot slices to their
    //  owners, the right call for a few slices that are ALL hot (four cores, p_1 = 58) and the wrong one for the few dozen hot
    //  slices of a skewed stream over 880: tb4z 218 -> 387 us, scripts/probes/r06_wide_hot_ab.sh; -1 keeps the column split)
    P.hdr[8 + 1] = wt[0] ? -1 : 0;
  }
}

template <int BITS, bool GRP>
__global__ __launch_bounds__(kWideThreads) void mbw_scatter_kernel(
    Dims d, int Nmax, const int* __restrict__ n_dev, const int64_t* __restrict__ indices,
    const int64_t* __restrict__ tableidx, const int64_t* __restrict__ rowidx, const int* __restrict__ cnt, Plan P,
    GrpArgs ga) {
  constexpr int BINS = 1 << BITS, K = BINS / kWideThreads;
  // 12 bits (round 5: 2048 < slices <= 4096 -- a two-core table of 11 M rows is p = [3317, 3317] -- took the multi-pass plan, nine
  // launches and 58 us at the benchmark's batch): sixteen per-wave rows of 4096 ints would be 256 KB, so the rows are 16-bit
  // and RELATIVE to the digit's first position of this work-group (a work-group spans 4096 positions), which lives in hbase.
  constexpr bool REL = BITS >= 12;
  using Cnt = typename std::conditional<REL, unsigned short, int>::type;
  extern __shared__ int wide_lds[];
  Cnt (*hrun)[BINS] = (Cnt (*)[BINS])wide_lds;  // [kWideWaves][BINS]
  int* wt = wide_lds + kWideWaves * BINS * (int)sizeof(Cnt) / (int)sizeof(int);  // [kWideWaves]
  int* hbase = wt + kWideWaves;                 // (REL) [BINS]
  __shared__ int gpos[GRP ? kMaxGroups + 1 : 1], gwg[GRP ? kMaxGroups + 1 : 1];
  const int t = blockIdx.y, tid = threadIdx.x, lane = lane_id(), w = tid / kWave;
  const int N = live_n(Nmax, n_dev);
  const CoreDec ct = core_dec(d, t);
  int beg = blockIdx.x * kWideSpan, end = min(N, beg + kWideSpan), sb = 0, pos0 = 0, wg0 = 0, wg1 = gridDim.x;
  int grp = 0, sg = d.S[t];  // group, its number of slices in this core
  if (GRP) {
    GrpMap m;
    if (!grp_map(d, ga, N, gpos, gwg, &m)) return;
    beg = m.beg; end = m.end; pos0 = m.pos0; wg0 = m.wg0; wg1 = m.wg1; grp = m.k;
    sb = slice_base(d, ct, t, m.tb0);
    sg = slice_base(d, ct, t, m.tb0 + ga.gsz) - sb;
  }
  {  // (wave-private row, zeroed 16 bytes per lane and round: a 12-bit row is 8 KB)
    int4* z = (int4*)&hrun[w][0];
    constexpr int kRowVec = BINS * (int)sizeof(Cnt) / 16;
    for (int e = lane; e < kRowVec; e += kWave) z[e] = make_int4(0, 0, 0, 0);
  }
  // this wave's kSB x 64 positions: key, peers of equal key in the batch, per-wave digit counts
  long long ix[kSB];
  int tb[kSB], kv[kSB], brow[kSB];
  unsigned long long peers[kSB];
  const int wbeg = beg + w * (kSB * kWave);
#pragma unroll
  for (int k = 0; k < kSB; ++k) {
    const int i = wbeg + k * kWave + lane;
    ix[k] = i < end ? indices[i] : 0;
    tb[k] = (i < end && tableidx) ? (int)tableidx[i] : 0;
    brow[k] = (t == 1 && rowidx && i < end) ? (int)rowidx[i] : 0;  // (the pivot's bag rows: fetched with the indices)
  }
  // thread owns digits tid * K .. + K - 1: totals over the (group's) work-groups, over the earlier ones (the
  // count rows of the previous launch: fetched here, behind the index loads, not after the barrier below)
  int tot[K], bef[K], dbase[K], sum = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) { tot[j] = 0; bef[j] = 0; }
  {
    const int* c = cnt + (size_t)t * gridDim.x * BINS + tid * K;
    for (int g0 = wg0; g0 < wg1; g0 += 8) {
      int v[8][K];
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < K; ++j) v[r][j] = (g0 + r < wg1) ? c[(size_t)(g0 + r) * BINS + j] : 0;
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          tot[j] += v[r][j];
          if (g0 + r < (int)blockIdx.x) bef[j] += v[r][j];
        }
    }
  }
#pragma unroll
  for (int k = 0; k < kSB; ++k) {
    const int i = wbeg + k * kWave + lane;
    const bool valid = i < end;
    kv[k] = valid ? min(max(slice_id(d, ct, t, tb[k], ix[k]) - sb, 0), BINS - 1) : 0;
    peers[k] = wave_match<BITS>((unsigned)kv[k], valid);
    if (valid && (peers[k] & lanemask_lt()) == 0) hrun[w][kv[k]] = (Cnt)(hrun[w][kv[k]] + __popcll(peers[k]));
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < K; ++j) sum += tot[j];
  const int inc = wave_incl_scan(sum);
  if (lane == kWave - 1) wt[w] = inc;
  __syncthreads();
  int wbase = 0;
  for (int k = 0; k < w; ++k) wbase += wt[k];
  dbase[0] = pos0 + wbase + inc - sum;  // first position of the thread's first digit
#pragma unroll
  for (int j = 1; j < K; ++j) dbase[j] = dbase[j - 1] + tot[j - 1];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int dg = tid * K + j;
    int b = REL ? 0 : dbase[j] + bef[j];
    if (REL) hbase[dg] = dbase[j] + bef[j];
#pragma unroll
    for (int k = 0; k < kWideWaves; ++k) { const int m = hrun[k][dg]; hrun[k][dg] = (Cnt)b; b += m; }
  }
  __syncthreads();
  if (!GRP) {
    if (blockIdx.x == 0) finish_wide<K>(d, t, tot, dbase, N, rowidx != nullptr, P, wt);
  } else if ((int)blockIdx.x == wg0) {  // the group's slice offsets, every core (the chunk list: mb_chunks_kernel)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int dg = tid * K + j;
      if (dg < sg) P.off[t][sb + dg] = dbase[j];
    }
    if (grp == ga.ngroups - 1 && tid == 0) P.off[t][d.S[t]] = N;
  }
#pragma unroll
  for (int k = 0; k < kSB; ++k) {
    const int i = wbeg + k * kWave + lane;
    const bool valid = i < end;
    if (valid) {
      const int before = hrun[w][kv[k]];
      const int pos = (REL ? hbase[kv[k]] : 0) + before + __popcll(peers[k] & lanemask_lt());
      if ((peers[k] & lanemask_lt()) == 0) hrun[w][kv[k]] = (Cnt)(before + __popcll(peers[k]));
      if (t != 1) {
        P.perm[t][pos] = i;
        P.ipos[t][i] = pos;
      } else {
        const int s0 = slice_id(d, core_dec(d, 0), 0, tb[k], ix[k]);
        const int s2 = d.T > 2 ? slice_id(d, core_dec(d, 2), 2, tb[k], ix[k]) : 0;
        const int s3 = d.T > 3 ? slice_id(d, core_dec(d, 3), 3, tb[k], ix[k]) : 0;
        P.lrec[pos] = make_int4(i, s0, s2, s3);
        if (rowidx) P.lrow[pos] = brow[k];
      }
    }
  }
}

// ---- multi-pass plan on full work-groups: more than 2048 slice ids, or more than kWideMaxG work-groups ----
// 8-bit passes like mb_count / mb_scatter, but spread as the wide-digit plan is: 1024-thread
// work-groups over 4096 positions of the current order (16 waves x kSB batches, all loads of a
// wave in flight together), one 256-bin count row per work-group, a column scan launch in
// between (mbp_scan: every scatter work-group reading all rows would be G^2 KiB of L2 traffic).
// The wave units of mb_count walk up to 4096 positions each on a quarter of the chip's wave slots.
struct MbpArgs {
  int N;               // lookups (upper bound when n_dev is set)
  const int* n_dev;
  int pass;
  int passes[TTX_MAX_CORES];
  int* cnt;            // [T][G][256]
};

__global__ __launch_bounds__(kWideThreads) void mbp_count_kernel(
    Dims d, MbpArgs A, const int64_t* __restrict__ indices, const int64_t* __restrict__ tableidx, Plan P) {
  __shared__ int hist[256];
  const int t = blockIdx.y, tid = threadIdx.x;
  if (A.pass >= A.passes[t]) return;
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  const int N = live_n(A.N, A.n_dev);
  int* key = P.sid[t];
  const int* src = (A.pass == 0) ? nullptr : ((A.pass & 1) ? P.scratch[t][1] : P.scratch[t][2]);
  const int shift = A.pass * 8;
  const CoreDec ct = core_dec(d, t);
  MbItem it[kSB];
#pragma unroll
  for (int k = 0; k < kSB; ++k) {
    const int i = blockIdx.x * kWideSpan + k * kWideThreads + tid;
    it[k] = mb_load(i, i < N, A.pass, d, t, ct, indices, tableidx, src, key);
  }
#pragma unroll
  for (int k = 0; k < kSB; ++k) {
    const int i = blockIdx.x * kWideSpan + k * kWideThreads + tid;
    if (i < N) {
      if (A.pass == 0 && A.passes[t] > 1) key[i] = it[k].kv;  // later passes chase order -> key
      atomicAdd(&hist[((unsigned)it[k].kv >> shift) & 255u], 1);
    }
  }
  __syncthreads();
  if (tid < 256) A.cnt[((size_t)t * gridDim.x + blockIdx.x) * 256 + tid] = hist[tid];
}

// counts -> first positions, in place, (digit major, work-group minor).  One work-group per core:
// thread = (digit, quarter of the rows); two walks over the rows with 8 loads in flight.
__global__ __launch_bounds__(kWideThreads) void mbp_scan_kernel(MbpArgs A, int G) {
  __shared__ int psum[4][256], dbase[256], wt[4];
  const int t = blockIdx.x, tid = threadIdx.x, dg = tid & 255, part = tid >> 8;
  if (A.pass >= A.passes[t]) return;
  const int per = (G + 3) / 4, r0 = min(G, part * per), r1 = min(G, r0 + per);
  int* c = A.cnt + (size_t)t * G * 256 + dg;
  int s = 0;
  for (int r = r0; r < r1; r += 8) {
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (r + j < r1) ? c[(size_t)(r + j) * 256] : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  psum[part][dg] = s;
  __syncthreads();
  if (tid < 256) {
    const int tot = psum[0][dg] + psum[1][dg] + psum[2][dg] + psum[3][dg];
    const int inc = wave_incl_scan(tot);
    if (lane_id() == kWave - 1) wt[tid / kWave] = inc;
    dbase[dg] = inc - tot;  // (+ the totals of the earlier waves, below)
  }
  __syncthreads();
  int run = dbase[dg];
  for (int k = 0; k < dg / kWave; ++k) run += wt[k];
  for (int k = 0; k < part; ++k) run += psum[k][dg];
  for (int r = r0; r < r1; r += 8) {
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (r + j < r1) ? c[(size_t)(r + j) * 256] : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (r + j < r1) c[(size_t)(r + j) * 256] = run;
      run += v[j];
    }
  }
}

__global__ __launch_bounds__(kWideThreads) void mbp_scatter_kernel(
    Dims d, MbpArgs A, const int64_t* __restrict__ indices, const int64_t* __restrict__ tableidx,
    const int64_t* __restrict__ rowidx, Plan P) {
  __shared__ int hrun[kWideWaves][256];
  const int t = blockIdx.y, tid = threadIdx.x, lane = lane_id(), w = tid / kWave;
  if (A.pass >= A.passes[t]) return;
  const int N = live_n(A.N, A.n_dev);
  int* key = P.sid[t];
  const int* src = (A.pass == 0) ? nullptr : ((A.pass & 1) ? P.scratch[t][1] : P.scratch[t][2]);
  const bool last = (A.pass == A.passes[t] - 1);
  int* dst = last ? P.perm[t] : ((A.pass & 1) ? P.scratch[t][2] : P.scratch[t][1]);
  int* sk = P.scratch[t][0];  // sorted keys (last pass): what mb_finish searches
  const int shift = A.pass * 8;
  const bool pivot = (t == 1);
  const CoreDec ct = core_dec(d, t);
  for (int e = lane; e < 256; e += kWave) hrun[w][e] = 0;  // (wave-private row)
  const int wbeg = blockIdx.x * kWideSpan + w * (kSB * kWave);
  MbItem it[kSB];
  unsigned long long peers[kSB];
#pragma unroll
  for (int k = 0; k < kSB; ++k) {
    const int i = wbeg + k * kWave + lane;
    it[k] = mb_load(i, i < N, A.pass, d, t, ct, indices, tableidx, src, key);
  }
  long long ix[kSB];
  int tb[kSB], brow[kSB];
  if (last && pivot) {  // the record gathers of the wave's positions, in flight together
#pragma unroll
    for (int k = 0; k < kSB; ++k) {
      const int i = wbeg + k * kWave + lane;
      const int v = it[k].val;
      ix[k] = i < N ? indices[v] : 0;
      tb[k] = (i < N && tableidx) ? (int)tableidx[v] : 0;
      brow[k] = (i < N && rowidx) ? (int)rowidx[v] : 0;
    }
  }
#pragma unroll
  for (int k = 0; k < kSB; ++k) {
    const int i = wbeg + k * kWave + lane;
    const bool valid = i < N;
    const unsigned dg = ((unsigned)it[k].kv >> shift) & 255u;
    peers[k] = wave_match8(dg, valid);
    if (valid && (peers[k] & lanemask_lt()) == 0) hrun[w][dg] += __popcll(peers[k]);
  }
  __syncthreads();
  if (tid < 256) {  // thread = digit: first position of (digit, this work-group), then of each wave
    int b = A.cnt[((size_t)t * gridDim.x + blockIdx.x) * 256 + tid];
#pragma unroll
    for (int k = 0; k < kWideWaves; ++k) { const int m = hrun[k][tid]; hrun[k][tid] = b; b += m; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kSB; ++k) {
    const int i = wbeg + k * kWave + lane;
    if (i < N) {
      const unsigned dg = ((unsigned)it[k].kv >> shift) & 255u;
      const int before = hrun[w][dg];
      const int pos = before + __popcll(peers[k] & lanemask_lt());
      if ((peers[k] & lanemask_lt()) == 0) hrun[w][dg] = before + __popcll(peers[k]);
      if (!(last && pivot)) dst[pos] = it[k].val;  // the pivot's final order lives in lrec.x
      if (last && !pivot) P.ipos[t][it[k].val] = pos;
      if (last) {
        sk[pos] = it[k].kv;
        if (pivot) {
          const int s0 = slice_id(d, core_dec(d, 0), 0, tb[k], ix[k]);
          const int s2 = d.T > 2 ? slice_id(d, core_dec(d, 2), 2, tb[k], ix[k]) : 0;
          const int s3 = d.T > 3 ? slice_id(d, core_dec(d, 3), 3, tb[k], ix[k]) : 0;
          P.lrec[pos] = make_int4(it[k].val, s0, s2, s3);
          if (rowidx) P.lrow[pos] = brow[k];
        }
      }
    }
  }
}

template <int BITS, bool GRP>
static int plan_build_wide(const Dims& d, int N, const int* n_dev, const int64_t* indices, const int64_t* tableidx,
                           const int64_t* rowidx, const Plan& P, hipStream_t stream, const GrpArgs& ga) {
  constexpr size_t lds = BITS >= 12 ? (size_t)kWideWaves * (1 << BITS) * sizeof(unsigned short) + (size_t)(kWideWaves + (1 << BITS)) * sizeof(int)
                                    : (size_t)(kWideWaves * (1 << BITS) + kWideWaves) * sizeof(int);
  if (lds > 64 * 1024) {
    const int rc_attr = allow_dynamic_lds((const void*)mbw_scatter_kernel<BITS, GRP>, (int)lds);
    if (rc_attr) return rc_attr;
  }
  const dim3 grid((N + kWideSpan - 1) / kWideSpan + (GRP ? ga.ngroups : 0), d.T);
  hipLaunchKernelGGL((mbw_count_kernel<BITS, GRP>), grid, dim3(kWideThreads), 0, stream, d, N, n_dev, indices, tableidx, P.cnt, ga);
  hipLaunchKernelGGL((mbw_scatter_kernel<BITS, GRP>), grid, dim3(kWideThreads), lds, stream, d, N, n_dev, indices, tableidx,
                     rowidx, (const int*)P.cnt, P, ga);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

// first position of the sorted keys with key >= s
__device__ __forceinline__ int lower_bound_key(const int* sk, int N, int s) {
  int lo = 0, hi = N;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sk[mid] < s) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// blockIdx.y = core: slice offsets of every core by binary search on its sorted keys
__global__ __launch_bounds__(1024) void mb_finish_kernel(Dims d, int Nmax, const int* __restrict__ n_dev, Plan P) {
  const int N = live_n(Nmax, n_dev);
  const int t = blockIdx.y, tid = threadIdx.x;
  const int* sk = P.scratch[t][0];
  const int S = d.S[t];
  for (int s = blockIdx.x * 1024 + tid; s <= S; s += gridDim.x * 1024) P.off[t][s] = (s == S) ? N : lower_bound_key(sk, N, s);
}

// ... then the pivot's chunk list from its offsets (one work-group; chunk slots are slice-major)
__global__ __launch_bounds__(1024) void mb_chunks_kernel(Dims d, int Nmax, const int* __restrict__ n_dev, int has_row, Plan P) {
  const int N = live_n(Nmax, n_dev);
  __shared__ int wtot[kPlanWaves + 1];
  const int tid = threadIdx.x;
  const int S1 = d.S[1], MC = P.MC;
  const int* off = P.off[1];
  int carry = 0;
  for (int s0 = 0; s0 < S1; s0 += 1024) {
    const int s = s0 + tid;
    int nch = 0, beg = 0, cnt = 0;
    if (s < S1) {
      beg = off[s];
      cnt = off[s + 1] - beg;
      nch = (cnt + MC - 1) / MC;
    }
    int total;
    const int ex = carry + block_excl_scan(nch, wtot, &total);
    if (s < S1) {
      P.chunk_off[s] = ex;
      for (int j = 0; j < nch; ++j) P.chunk_rec[ex + j] = make_int4(s, beg + j * MC, min(MC, cnt - j * MC), ex + j);
    }
    carry += total;
  }
  for (int c = carry + tid; c < P.max_chunks; c += 1024) P.chunk_rec[c] = make_int4(0, 0, 0, 0);
  // (round 6) hot slices per core for reduce_apply, as the one-launch routes leave them (finish_single_pass / finish_wide): the
  // thin cores' counts from their offset tables (complete: written by an earlier launch), the pivot's as none / some.  Every
  // thread counts its own slices with eight offset pairs in flight, ONE work-group sum per core at the end (integer: order-free)
  // -- a barrier-bounded round per 1024 slices was a trip to memory each: 21 -> 39 us for this kernel at 26 tables.
  __shared__ int hs[TTX_MAX_CORES + 1];
  if (tid <= TTX_MAX_CORES) hs[tid] = 0;
  int loc[TTX_MAX_CORES], anyp = 0;
#pragma unroll
  for (int t = 0; t < TTX_MAX_CORES; ++t) {
    loc[t] = 0;
    if (t == 1 || t >= d.T) continue;
    const int* offt = P.off[t];
    const int St = d.S[t];
    for (int s0 = tid; s0 < St; s0 += 8 * 1024) {
      int len[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = min(s0 + u * 1024, St - 1);
        len[u] = offt[s + 1] - offt[s];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) loc[t] += (s0 + u * 1024 < St && len[u] > 2 * kSegThin) ? 1 : 0;
    }
  }
  for (int s = tid; s < S1; s += 1024) anyp |= ((off[s + 1] - off[s] + MC - 1) / MC > kHotRowsPivot) ? 1 : 0;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TTX_MAX_CORES; ++t)
    if (loc[t]) atomicAdd(&hs[t], loc[t]);
  if (anyp) atomicOr(&hs[TTX_MAX_CORES], 1);
  __syncthreads();
  int nhot[TTX_MAX_CORES];
#pragma unroll
  for (int t = 0; t < TTX_MAX_CORES; ++t) nhot[t] = (t == 1) ? (hs[TTX_MAX_CORES] ? -1 : 0) : (t < d.T ? hs[t] : -1);
  if (tid == 0) {
    P.chunk_off[S1] = carry;
    P.hdr[0] = carry;
    P.hdr[1] = MC;
    P.hdr[2] = N;
    P.hdr[3] = has_row;
    P.hdr[kHdrT4Valid] = 0;
#pragma unroll
    for (int tt = 0; tt < TTX_MAX_CORES; ++tt) P.hdr[8 + tt] = nhot[tt];
  }
}

static void launch_finish(const Dims& d, int N, const int* n_dev, bool has_row, const Plan& P, hipStream_t stream) {
  int smax = 1;
  for (int t = 0; t < d.T; ++t) if (d.S[t] + 1 > smax) smax = d.S[t] + 1;
  hipLaunchKernelGGL(mb_finish_kernel, dim3((smax + 1023) / 1024, d.T), dim3(1024), 0, stream, d, N, n_dev, P);
  hipLaunchKernelGGL(mb_chunks_kernel, dim3(1), dim3(1024), 0, stream, d, N, n_dev, has_row ? 1 : 0, P);
}

// would the plan of N lookups use table groups if it were given table-major offsets?  (plan_build_mb's own order of choices)
bool plan_groups_tables(const Dims& d, long long N) {
  if (d.num_tables <= 1 || d.tab) return false;
  int smax = 1;
  for (int t = 0; t < d.T; ++t) if (d.S[t] > smax) smax = d.S[t];
  if (smax <= 256) return false;                                                    // one 8-bit pass
  if (smax <= 4096 && (N + kWideSpan - 1) / kWideSpan <= kWideMaxG) return false;   // one wide digit, no groups
  return true;
}

static int plan_build_mb(const Dims& d, int N, const int* n_dev, const int64_t* indices, const int64_t* tableidx,
                         const int64_t* rowidx, const Plan& P, hipStream_t stream, const int64_t* offsets,
                         int bags_per_table) {
  MbArgs A;
  A.N = N;
  A.n_dev = n_dev;
  int maxp = 1, passes[TTX_MAX_CORES];  // 8-bit passes the slice ids of each core need
  for (int t = 0; t < TTX_MAX_CORES; ++t) {
    passes[t] = 0;
    if (t < d.T) {
      int bits = 0;
      while ((1ll << bits) < d.S[t]) ++bits;
      passes[t] = (bits + 7) / 8 > 0 ? (bits + 7) / 8 : 1;
      if (passes[t] > maxp) maxp = passes[t];
    }
  }
  if (maxp == 1 && N <= kOneMaxN && !d.tab) {
    hipLaunchKernelGGL(mb_single_kernel<false>, dim3((N + kOneWaves * kOneUnit - 1) / (kOneWaves * kOneUnit) + TTX_PLAN_XWG, d.T),
                       dim3(kOneThreads), 0, stream, d, N, n_dev, indices, tableidx, rowidx, P, Prologue{}, ProBatch{});
    TTX_HIP(hipGetLastError());
    return TTX_OK;
  }
  // (tables of different row factors, d.tab: only the wide-digit and the multi-pass plan decode them)
  // (finish_wide packs a slice's full-chunk count into the upper 18 bits of an int32 prefix sum: the route is for fewer than 2^18
  //  full chunks -- always, unless the generic kernels' last-resort tile or the test knob leaves MC = 1 with more than 262,144
  //  lookups; those take the passes below)
  if ((maxp > 1 || d.tab) && (N + kWideSpan - 1) / kWideSpan <= kWideMaxG && N / (P.MC > 0 ? P.MC : 1) < (1 << 18)) {  // one wide digit instead of two passes?
    int smax = 1;
    for (int t = 0; t < d.T; ++t) if (d.S[t] > smax) smax = d.S[t];
    if (smax <= 1024) return plan_build_wide<10, false>(d, N, n_dev, indices, tableidx, rowidx, P, stream, GrpArgs{});
    if (smax <= 2048) return plan_build_wide<11, false>(d, N, n_dev, indices, tableidx, rowidx, P, stream, GrpArgs{});
    if (smax <= 4096) return plan_build_wide<12, false>(d, N, n_dev, indices, tableidx, rowidx, P, stream, GrpArgs{});
  }
  if (offsets && d.num_tables > 1 && maxp > 1) {
    // more slice ids than one digit holds, bags known to be table-major (the module's offsets): table groups.
    // As many groups as allowed -- every scatter work-group reads the count rows of its whole group
    int pmax = 1;
    for (int t = 0; t < d.T; ++t) if (d.p[t] > pmax) pmax = d.p[t];  // (the largest table's, if they differ)
    int gsz = (d.num_tables + kMaxGroups - 1) / kMaxGroups;
    const int ngroups = (d.num_tables + gsz - 1) / gsz;
    const long long rows_per_group = (long long)N / kWideSpan / ngroups + 1;
    if ((long long)gsz * pmax <= 2048 && rows_per_group <= kWideMaxG) {
      GrpArgs ga{offsets, bags_per_table, gsz, ngroups};
      const int rc = (long long)gsz * pmax <= 1024
                         ? plan_build_wide<10, true>(d, N, n_dev, indices, tableidx, rowidx, P, stream, ga)
                         : plan_build_wide<11, true>(d, N, n_dev, indices, tableidx, rowidx, P, stream, ga);
      if (rc != TTX_OK) return rc;
      hipLaunchKernelGGL(mb_chunks_kernel, dim3(1), dim3(1024), 0, stream, d, N, n_dev, rowidx ? 1 : 0, P);
      TTX_HIP(hipGetLastError());
      return TTX_OK;
    }
  }
  if (maxp > 1 || N > kMbFuseU * 4096 || d.tab) {  // 8-bit passes on full work-groups, then mb_finish
    MbpArgs B;
    B.N = N;
    B.n_dev = n_dev;
    B.cnt = P.cnt;
    for (int t = 0; t < TTX_MAX_CORES; ++t) B.passes[t] = passes[t];
    const int G = (N + kWideSpan - 1) / kWideSpan;
    const dim3 grid(G, d.T);
    for (int ps = 0; ps < maxp; ++ps) {
      B.pass = ps;
      hipLaunchKernelGGL(mbp_count_kernel, grid, dim3(kWideThreads), 0, stream, d, B, indices, tableidx, P);
      hipLaunchKernelGGL(mbp_scan_kernel, dim3(d.T), dim3(kWideThreads), 0, stream, B, G);
      hipLaunchKernelGGL(mbp_scatter_kernel, grid, dim3(kWideThreads), 0, stream, d, B, indices, tableidx, rowidx, P);
    }
    launch_finish(d, N, n_dev, rowidx != nullptr, P, stream);
    TTX_HIP(hipGetLastError());
    return TTX_OK;
  }
  // one 8-bit pass, up to ~1 M lookups: wave units of 256..4096 positions (<= kMbFuseU of them), the
  // scatter launch scans the unit counts itself and its first work-group writes offsets / chunk list
  A.unit = 256;
  if ((N + 255) / 256 > kMbFuseU) A.unit = ((N + kMbFuseU - 1) / kMbFuseU + 63) / 64 * 64;
  A.U = (N + A.unit - 1) / A.unit;
  A.cnt = P.cnt;
  const dim3 gu((A.U + kMbUnits - 1) / kMbUnits, d.T);
  hipLaunchKernelGGL(mb_count_kernel, gu, dim3(kMbThreads), 0, stream, d, A, indices, tableidx);
  hipLaunchKernelGGL(mb_scatter_kernel, gu, dim3(kMbThreads), 0, stream, d, A, indices, tableidx, rowidx, P);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

// ---- opt-in index range check (TTX_CHECK_INDICES=1) ----------------------------------------------------------------------
// The reference decodes any int64 it is handed (i_0 = idx / L_0 with no bound: an index >= prod(p) reads past core 0,
// tt_embeddings_cuda.cu:795-799); the plan kernels here CLAMP every factor into its range instead, so a bad index silently
// becomes another row.  With TTX_CHECK_INDICES=1 every entry point that reads a batch's indices first verifies 0 <= idx < prod(p)
// of the lookup's table (and 0 <= table < num_tables) and returns TTX_EINVAL naming the first offender.  One small launch plus a
// host read-back per batch: a debugging aid -- off by default, skipped (with the clamp as the behaviour) while the stream is
// being captured.
__global__ __launch_bounds__(256) void check_indices_kernel(Dims d, int Nmax, const int* __restrict__ n_dev,
                                                            const int64_t* __restrict__ indices,
                                                            const int64_t* __restrict__ tableidx, long long* __restrict__ bad) {
  const int N = live_n(Nmax, n_dev);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const long long tb = tableidx ? tableidx[i] : 0;
    bool ok = tb >= 0 && tb < d.num_tables;
    if (ok) {
      const long long E = d.tab ? d.tab->L[tb][0] * d.tab->p[tb][0] : d.L[0] * d.p[0];
      ok = indices[i] >= 0 && indices[i] < E;
    }
    if (!ok) {
      atomicAdd((unsigned long long*)&bad[0], 1ull);
      atomicMin((unsigned long long*)&bad[1], (unsigned long long)i);
    }
  }
}
bool check_indices_on() {
  static const bool on = getenv("TTX_CHECK_INDICES") && atoi(getenv("TTX_CHECK_INDICES")) != 0;
  return on;
}
int check_indices(const Dims& d, long long nnz, const int* n_dev, const int64_t* indices, const int64_t* tableidx, hipStream_t stream) {
  if (!check_indices_on() || nnz <= 0 || !indices) return TTX_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return TTX_OK; }
  if (cs != hipStreamCaptureStatusNone) return TTX_OK;  // (a read-back cannot be captured: the plan's clamp applies)
  long long* bad = nullptr;
  TTX_HIP(hipMalloc((void**)&bad, 2 * sizeof(long long)));  // (a debugging path: the synchronising allocator is fine)
  const long long init[2] = {0, 0x7fffffffffffffffll};
  TTX_HIP(hipMemcpyAsync(bad, init, sizeof(init), hipMemcpyHostToDevice, stream));
  const int blocks = (int)((nnz + 255) / 256 < 1024 ? (nnz + 255) / 256 : 1024);
  hipLaunchKernelGGL(check_indices_kernel, dim3(blocks), dim3(256), 0, stream, d, (int)nnz, n_dev, indices, tableidx, bad);
  long long got[2] = {0, 0};
  TTX_HIP(hipMemcpyAsync(got, bad, sizeof(got), hipMemcpyDeviceToHost, stream));
  TTX_HIP(hipStreamSynchronize(stream));
  (void)hipFree(bad);
  if (got[0] == 0) return TTX_OK;
  long long idx = 0, tb = 0;
  (void)hipMemcpy(&idx, indices + got[1], sizeof(idx), hipMemcpyDeviceToHost);
  if (tableidx) (void)hipMemcpy(&tb, tableidx + got[1], sizeof(tb), hipMemcpyDeviceToHost);
  const long long E = (d.tab && tb >= 0 && tb < d.num_tables) ? -1 : d.L[0] * d.p[0];
  TTX_FAIL(TTX_EINVAL, "TTX_CHECK_INDICES: %lld of %lld lookups are out of range; the first is lookup %lld: index %lld of table %lld "
           "(valid: 0 <= index < %s%lld, 0 <= table < %d)", got[0], nnz, got[1], idx, tb, E < 0 ? "prod(p) of its table; shown: " : "",
           E < 0 ? -1ll : E, d.num_tables);
}

int plan_build(const Dims& d, long long nnz, const int64_t* indices,
               const int64_t* tableidx, const int64_t* rowidx, const Plan& P, hipStream_t stream,
               const int* n_dev, const int64_t* offsets, int bags_per_table) {
  if (nnz < 0 || nnz >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "nnz=%lld out of range", nnz);
  {
    const int rc_chk = check_indices(d, nnz, n_dev, indices, tableidx, stream);
    if (rc_chk) return rc_chk;
  }
  if (P.MC <= 0) TTX_FAIL(TTX_EUNSUPPORTED, "TT shape does not fit the LDS of any kernel variant (core-1 slice %d x %d floats)", d.k[0], d.n[0]);
  ProfScope ps(TTX_PROF_PLAN, stream);
  if (nnz > 1024 || !d.idx32 || n_dev)
    return plan_build_mb(d, (int)nnz, n_dev, indices, tableidx, rowidx, P, stream, offsets, bags_per_table);
  {  // tiny batch: one launch, one work-group per core, everything on chip
    const size_t lds = (256 * kPlanWaves + 32 + 2 * ((nnz + 63) / 64 * 64)) * sizeof(int);
    const int per = (((int)nnz + kPlanWaves - 1) / kPlanWaves + kWave - 1) / kWave * kWave;
    const int nb = per / kWave;
#define TTX_PLAN_LAUNCH(BPW)                                                                          \
  do {                                                                                                \
    const int rc_attr = allow_dynamic_lds((const void*)plan_small_kernel<BPW>, 160 * 1024);          \
    if (rc_attr) return rc_attr;                                                                      \
    hipLaunchKernelGGL(plan_small_kernel<BPW>, dim3(d.T), dim3(kPlanThreads), lds, stream, d,         \
                       (int)nnz, indices, tableidx, rowidx, P, debug_stamps());                               \
  } while (0)
    if (nb <= 2) TTX_PLAN_LAUNCH(2);
    else if (nb <= 4) TTX_PLAN_LAUNCH(4);
    else if (nb <= 8) TTX_PLAN_LAUNCH(8);
    else if (nb <= 12) TTX_PLAN_LAUNCH(12);
    else TTX_PLAN_LAUNCH(16);
#undef TTX_PLAN_LAUNCH
  }
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

// The plans of `nbatch` batches of one size in ONE launch (grid.z = batch): batch z reads indices / tableidx / rowidx at
// z * nnz elements, its live count at n_dev[z], and writes its plan z * plan_stride bytes behind `plans`.  Only the
// single-launch plan does this (plan_batches_ok); same plans as plan_build(.., n_dev + z) batch by batch.
bool plan_batches_ok(const Dims& d, long long nnz) {
  if (nnz > kOneMaxN || d.tab) return false;
  for (int t = 0; t < d.T; ++t) if (d.S[t] > 256) return false;
  return true;
}

int plan_build_batches(const Dims& d, int nbatch, long long nnz, const int* n_dev, const int64_t* indices,
                       const int64_t* tableidx, const int64_t* rowidx, void* plans, size_t plan_stride,
                       hipStream_t stream) {
  if (!plan_batches_ok(d, nnz)) TTX_FAIL(TTX_EINVAL, "plan_build_batches: batch shape needs the multi-launch plans");
  for (int z = 0; z < nbatch && check_indices_on(); ++z) {
    const int rc_chk = check_indices(d, nnz, n_dev ? n_dev + z : nullptr, indices + (size_t)z * nnz, tableidx ? tableidx + (size_t)z * nnz : nullptr, stream);
    if (rc_chk) return rc_chk;
  }
  ProfScope ps(TTX_PROF_PLAN, stream);
  const Plan P = carve_plan(d, nnz, plans);
  ProBatch mb{};
  mb.out_stride = nnz;
  mb.plan_stride = (long long)plan_stride;
  const int N = (int)nnz;
  hipLaunchKernelGGL(mb_single_kernel<false>,
                     dim3((N + kOneWaves * kOneUnit - 1) / (kOneWaves * kOneUnit) + TTX_PLAN_XWG, d.T, nbatch),
                     dim3(kOneThreads), 0, stream, d, N, n_dev, indices, tableidx, rowidx, P, Prologue{}, mb);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

// ---- duplicate lookups (DedupMap, ttx_internal.h) ----------------------------------------------------------
// One work-group sorts the batch's (table, index) keys -- 32 bits, stable LSD radix sort in LDS, the same wave
// ranking as plan_small_kernel -- so that equal pairs end up adjacent with their occurrences in index order;
// run heads are the distinct pairs.  Deterministic (no ordering-dependent atomics), one launch, <= 16384 lookups
// (144 KB of LDS); larger batches and key spaces beyond 2^32 are not deduplicated (same results either way:
// sharing the contraction of duplicates never changes a value, only how often it is computed).
// (batches beyond the single-work-group map also keep the key sort's buffers here: keys, values, the sort's workspace,
//  run heads per block of 1024 sorted positions)
// header: [0] distinct pairs, then (8-byte aligned, from int 16 on) tstart[kDedupMaxTables + 1]: first pair of every table
constexpr int kDedupMaxTables = 1023;
constexpr size_t kDedupHdrBytes = 64 + (kDedupMaxTables + 1) * sizeof(int64_t);
static size_t dedup_map_bytes(long long nnz) {
  const size_t n = r64((size_t)nnz + 1);
  return align_up(kDedupHdrBytes) + 3 * align_up(n * sizeof(int64_t)) + 3 * align_up(n * sizeof(int));
}
static size_t dedup_blocks(long long nnz) { return ((size_t)nnz + 1023) / 1024; }
size_t dedup_bytes(long long nnz) {
  return dedup_map_bytes(nnz) + 2 * align_up((size_t)nnz * 8) + align_up(sort_pairs_ws_bytes(nnz)) +
         align_up((dedup_blocks(nnz) + 1) * sizeof(int));
}

DedupMap carve_dedup(long long nnz, void* base) {
  const size_t n = r64((size_t)nnz + 1);
  char* cur = (char*)base;
  auto take = [&](size_t bytes) { char* r = cur; cur += align_up(bytes); return r; };
  DedupMap M;
  M.nu = (int*)take(kDedupHdrBytes);
  M.uidx = (int64_t*)take(n * sizeof(int64_t));
  M.utab = (int64_t*)take(n * sizeof(int64_t));
  M.iota = (int64_t*)take(n * sizeof(int64_t));
  M.uid = (int*)take(n * sizeof(int));
  M.occ = (int*)take(n * sizeof(int));
  M.occ_off = (int*)take(n * sizeof(int));
  return M;
}

static unsigned long long dedup_key_space(const Dims& d) {  // tables * prod(p), or 0 if beyond 2^32
  if (d.tab || !d.idx32) return 0;
  unsigned long long e = 1;
  for (int t = 0; t < d.T; ++t) e *= (unsigned long long)d.p[t];
  const unsigned long long all = e * (unsigned long long)d.num_tables;
  return (e <= (1ull << 32) && all <= (1ull << 32)) ? all : 0;
}

// tables * prod(p) as a 64-bit key space (batches beyond the single-work-group map sort 64-bit keys), or 0
static unsigned long long dedup_key_space64(const Dims& d) {
  if (d.tab) return 0;
  unsigned long long e = 1;
  for (int t = 0; t < d.T; ++t) {
    if (e > (1ull << 61) / (unsigned long long)d.p[t]) return 0;
    e *= (unsigned long long)d.p[t];
  }
  if (e > (1ull << 61) / (unsigned long long)d.num_tables) return 0;
  return e * (unsigned long long)d.num_tables;
}

bool dedup_supported(const Dims& d, long long nnz) {
  if (nnz <= 0 || nnz >= (1ll << 31)) return false;
  if (nnz <= kDedupMaxN && dedup_key_space(d) != 0) return true;   // one work-group sorts the batch in LDS
  return dedup_key_space64(d) != 0;                                // multi-work-group 64-bit key sort
}

template <int kBPW>
__global__ __launch_bounds__(kPlanThreads) void dedup_small_kernel(int N, unsigned long long E, int num_tables, int passes,
                                                                   const int64_t* __restrict__ indices,
                                                                   const int64_t* __restrict__ tableidx, DedupMap M) {
  extern __shared__ __attribute__((aligned(16))) int lds[];
  int* hist = lds;                        // [256][kPlanWaves]
  int* wtot = hist + 256 * kPlanWaves;    // [kPlanWaves + 1] (+pad)
  int* keyL = wtot + 32;                  // [N]
  int* valL = keyL + ((N + 63) / 64 * 64);  // [N]
  const int tid = threadIdx.x, lane = lane_id(), w = tid / kWave;
  const int per = ((N + kPlanWaves - 1) / kPlanWaves + kWave - 1) / kWave * kWave;
  const int nb = per / kWave;  // <= kBPW
  const int wbeg = w * per, wend = min(N, wbeg + per);
  unsigned k[kBPW];
  int v[kBPW], r[kBPW];
  {
    const int last_i = N - 1;
    long long ix[kBPW];
    int tbv[kBPW];
#pragma unroll
    for (int b = 0; b < kBPW; ++b) {
      const int i = min(wbeg + b * kWave + lane, last_i);
      ix[b] = indices[i];
      tbv[b] = (tableidx && num_tables > 1) ? (int)tableidx[i] : 0;
    }
#pragma unroll
    for (int b = 0; b < kBPW; ++b) {
      // out-of-range inputs are clamped into the table's key range (the plan's decode clamps them likewise)
      const unsigned long long e = (unsigned long long)max(ix[b], 0ll);
      const unsigned long long tb = (unsigned long long)min(max(tbv[b], 0), num_tables - 1);
      k[b] = (unsigned)(tb * E + (e < E ? e : E - 1));
      v[b] = wbeg + b * kWave + lane;
    }
  }
  for (int ps = 0; ps < passes; ++ps) {
    const int shift = ps * 8;
    for (int e = tid; e < 256 * kPlanWaves; e += kPlanThreads) hist[e] = 0;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < kBPW; ++b) {
      if (b < nb) {  // wave-uniform
        const bool valid = wbeg + b * kWave + lane < wend;
        const unsigned dg = (k[b] >> shift) & 255u;
        const unsigned long long peers = wave_match8(dg, valid);
        if (valid) {
          const int before = hist[dg * kPlanWaves + w];
          r[b] = before + __popcll(peers & lanemask_lt());
          if ((peers & lanemask_lt()) == 0) hist[dg * kPlanWaves + w] = before + __popcll(peers);
        }
      }
    }
    __syncthreads();
    {
      int v0 = hist[tid * 4 + 0], v1 = hist[tid * 4 + 1], v2 = hist[tid * 4 + 2], v3 = hist[tid * 4 + 3];
      int total;
      int ex = block_excl_scan(v0 + v1 + v2 + v3, wtot, &total);
      hist[tid * 4 + 0] = ex;
      hist[tid * 4 + 1] = ex + v0;
      hist[tid * 4 + 2] = ex + v0 + v1;
      hist[tid * 4 + 3] = ex + v0 + v1 + v2;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < kBPW; ++b) {
      if (b < nb && wbeg + b * kWave + lane < wend) {
        const unsigned dg = (k[b] >> shift) & 255u;
        const int pos = hist[dg * kPlanWaves + w] + r[b];
        keyL[pos] = (int)k[b];
        valL[pos] = v[b];
      }
    }
    __syncthreads();
    if (ps + 1 < passes) {
#pragma unroll
      for (int b = 0; b < kBPW; ++b) {
        const int i = wbeg + b * kWave + lane;
        if (b < nb && i < wend) { k[b] = (unsigned)keyL[i]; v[b] = valL[i]; }
      }
      __syncthreads();
    }
  }
  // run heads -> distinct pairs.  Thread tid owns positions [tid*cpt, (tid+1)*cpt) of the sorted order.
  const int cpt = (N + kPlanThreads - 1) / kPlanThreads;
  const int beg = min(N, tid * cpt), end = min(N, beg + cpt);
  int cnt = 0;
  for (int i = beg; i < end; ++i) cnt += (i == 0 || keyL[i] != keyL[i - 1]) ? 1 : 0;
  int total;
  int u = block_excl_scan(cnt, wtot, &total) - 1;  // pair of position beg - 1
  for (int i = beg; i < end; ++i) {
    const unsigned key = (unsigned)keyL[i];
    if (i == 0 || keyL[i] != keyL[i - 1]) {
      ++u;
      const unsigned long long tb = (unsigned long long)key / E;
      M.occ_off[u] = i;
      M.uidx[u] = (int64_t)((unsigned long long)key - tb * E);
      M.utab[u] = (int64_t)tb;
    }
    const int n = valL[i];
    M.uid[n] = u;
    M.occ[i] = n;
    M.iota[i] = i;
  }
  if (tid == 0) {
    M.nu[0] = total;
    M.occ_off[total] = N;
  }
}

// ---- the same map for ANY batch size (round 3): the batch's 64-bit keys table * prod(p) + index go through the
// multi-work-group stable radix sort that cache_populate uses (ttx_cache.hip sort_pairs_desc: descending, so the
// distinct pairs come out in descending key order -- the order does not matter to anybody --, occurrences of a pair in
// index order because the sort is stable); run heads are counted per block of 1024 sorted positions, scanned by one
// work-group, and a last launch writes the map.  No ordering-dependent atomics: deterministic.  The number of 8-bit passes
// follows from the geometry (bytes of tables * prod(p)), so nothing is read back and the build captures into a hipGraph.
constexpr int kDdThreads = 1024;
__global__ __launch_bounds__(kDdThreads) void dd_keys_kernel(int N, unsigned long long E, int num_tables,
                                                            const int64_t* __restrict__ indices,
                                                            const int64_t* __restrict__ tableidx, int64_t* __restrict__ keys,
                                                            int64_t* __restrict__ vals) {
  const int i = blockIdx.x * kDdThreads + threadIdx.x;
  if (i >= N) return;
  // out-of-range inputs are clamped into the table's key range (the plan's decode clamps them likewise)
  const unsigned long long e = (unsigned long long)max(indices[i], (int64_t)0);
  const int tbv = (tableidx && num_tables > 1) ? (int)tableidx[i] : 0;
  const unsigned long long tb = (unsigned long long)min(max(tbv, 0), num_tables - 1);
  // complemented: the (descending, stable) pair sort then leaves the pairs ASCENDING by (table, index) -- table-major like
  // the module's bags, which is what lets the plan of the pairs sort each table group by itself (dedup_tstart_kernel)
  keys[i] = (int64_t)((unsigned long long)num_tables * E - 1ull - (tb * E + (e < E ? e : E - 1)));
  vals[i] = i;
}

// run heads among this block's 1024 sorted positions -> *total; returns this thread's inclusive count
__device__ __forceinline__ int dd_block_heads(bool head, int* wt, int* total) {
  const int lane = lane_id(), w = threadIdx.x / kWave;
  const unsigned long long hm = __ballot(head);
  const int inc_w = __popcll(hm & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull)));
  if (lane == 0) wt[w] = __popcll(hm);
  __syncthreads();
  int base = 0, tot = 0;
  for (int k = 0; k < kDdThreads / kWave; ++k) {
    const int c = wt[k];
    if (k < w) base += c;
    tot += c;
  }
  *total = tot;
  __syncthreads();
  return base + inc_w;
}

__global__ __launch_bounds__(kDdThreads) void dd_count_kernel(int N, const int64_t* __restrict__ sk, int* __restrict__ blk_cnt) {
  __shared__ int wt[kDdThreads / kWave];
  const int i = blockIdx.x * kDdThreads + threadIdx.x;
  const bool head = i < N && (i == 0 || sk[i] != sk[i - 1]);
  int total;
  dd_block_heads(head, wt, &total);
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = total;
}

// exclusive scan of the per-block head counts (one work-group); the total is the number of distinct pairs
__global__ __launch_bounds__(kDdThreads) void dd_scan_kernel(int nblk, int N, int* __restrict__ blk_cnt, DedupMap M) {
  __shared__ int wt[kDdThreads / kWave + 1];
  int carry = 0;
  for (int b0 = 0; b0 < nblk; b0 += kDdThreads) {
    const int i = b0 + threadIdx.x;
    const int v = i < nblk ? blk_cnt[i] : 0;
    const int inc = wave_incl_scan(v);
    const int w = threadIdx.x / kWave;
    if (lane_id() == kWave - 1) wt[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
      int run = 0;
      for (int k = 0; k < kDdThreads / kWave; ++k) { const int c = wt[k]; wt[k] = run; run += c; }
      wt[kDdThreads / kWave] = run;
    }
    __syncthreads();
    if (i < nblk) blk_cnt[i] = carry + wt[w] + inc - v;
    carry += wt[kDdThreads / kWave];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    M.nu[0] = carry;
    M.occ_off[carry] = N;
  }
}

__global__ __launch_bounds__(kDdThreads) void dd_emit_kernel(int N, unsigned long long E, unsigned long long all, const int64_t* __restrict__ sk,
                                                            const int64_t* __restrict__ sv, const int* __restrict__ blk_base,
                                                            DedupMap M) {
  __shared__ int wt[kDdThreads / kWave];
  const int i = blockIdx.x * kDdThreads + threadIdx.x;
  const int64_t key = i < N ? sk[i] : 0;
  const bool head = i < N && (i == 0 || key != sk[i - 1]);
  int total;
  const int inc = dd_block_heads(head, wt, &total);
  if (i >= N) return;
  const int u = blk_base[blockIdx.x] + inc - 1;  // the pair of position i: heads at or before it, minus one
  const int n = (int)sv[i];
  M.uid[n] = u;
  M.occ[i] = n;
  M.iota[i] = i;
  if (head) {
    const unsigned long long real = all - 1ull - (unsigned long long)key;  // (dd_keys_kernel stores the complement)
    const unsigned long long tb = real / E;
    M.occ_off[u] = i;
    M.uidx[u] = (int64_t)(real - tb * E);
    M.utab[u] = (int64_t)tb;
  }
}

// first pair of every table in the (ascending, table-major) pair list: tstart[t] = lower bound of t in utab[0, nu)
__global__ __launch_bounds__(256) void dedup_tstart_kernel(int num_tables, DedupMap M) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t > num_tables) return;
  const int nu = M.nu[0];
  int lo = 0, hi = nu;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (M.utab[mid] < (int64_t)t) lo = mid + 1; else hi = mid;
  }
  dedup_tstart(M)[t] = lo;
}

static int dedup_build_large(const Dims& d, long long nnz, const int64_t* indices, const int64_t* tableidx, const DedupMap& M,
                             hipStream_t stream) {
  const unsigned long long all = dedup_key_space64(d);
  const unsigned long long E = all / (unsigned long long)d.num_tables;
  int bits = 1;
  while (bits < 62 && (1ull << bits) < all) ++bits;
  const int passes = (bits + 7) / 8;
  const int N = (int)nnz;
  char* extra = (char*)M.nu + dedup_map_bytes(nnz);
  int64_t* keys = (int64_t*)extra;
  int64_t* vals = (int64_t*)(extra + align_up((size_t)nnz * 8));
  char* sws = extra + 2 * align_up((size_t)nnz * 8);
  int* blk = (int*)(sws + align_up(sort_pairs_ws_bytes(nnz)));
  const int nblk = (int)dedup_blocks(nnz);
  ProfScope ps(TTX_PROF_PLAN, stream);
  hipLaunchKernelGGL(dd_keys_kernel, dim3(nblk), dim3(kDdThreads), 0, stream, N, E, d.num_tables, indices, tableidx, keys, vals);
  int64_t *sk = nullptr, *sv = nullptr;
  const int rc = sort_pairs_desc(nnz, keys, vals, sws, &sk, &sv, stream, passes);
  if (rc) return rc;
  hipLaunchKernelGGL(dd_count_kernel, dim3(nblk), dim3(kDdThreads), 0, stream, N, sk, blk);
  hipLaunchKernelGGL(dd_scan_kernel, dim3(1), dim3(kDdThreads), 0, stream, nblk, N, blk, M);
  hipLaunchKernelGGL(dd_emit_kernel, dim3(nblk), dim3(kDdThreads), 0, stream, N, E, all, sk, sv, blk, M);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

// The same map over 64-bit keys SOMEBODY ELSE wrote (round 6, ttx_cache.hip: the cached lookups of a batch keyed by their cache row,
// for the atomic-free cache-row update): the caller fills dedup_key_buffers()'s keys[i] = all - 1 - key_i (complemented, like
// dd_keys_kernel) and vals[i] = i for i < nnz, with every key_i < all; the map then lists the distinct keys ascending in uidx
// (utab = 0), every key's occurrences in index order.  M must have been carved from dedup_bytes(nnz) bytes.
void dedup_key_buffers(const DedupMap& M, long long nnz, int64_t** keys, int64_t** vals) {
  char* extra = (char*)M.nu + dedup_map_bytes(nnz);
  *keys = (int64_t*)extra;
  *vals = (int64_t*)(extra + align_up((size_t)nnz * 8));
}
int dedup_build_from_keys(long long nnz, unsigned long long all, const DedupMap& M, hipStream_t stream) {
  int bits = 1;
  while (bits < 62 && (1ull << bits) < all) ++bits;
  const int passes = (bits + 7) / 8;
  const int N = (int)nnz;
  int64_t *keys, *vals;
  dedup_key_buffers(M, nnz, &keys, &vals);
  char* sws = (char*)vals + align_up((size_t)nnz * 8);
  int* blk = (int*)(sws + align_up(sort_pairs_ws_bytes(nnz)));
  const int nblk = (int)dedup_blocks(nnz);
  int64_t *sk = nullptr, *sv = nullptr;
  const int rc = sort_pairs_desc(nnz, keys, vals, sws, &sk, &sv, stream, passes);
  if (rc) return rc;
  hipLaunchKernelGGL(dd_count_kernel, dim3(nblk), dim3(kDdThreads), 0, stream, N, sk, blk);
  hipLaunchKernelGGL(dd_scan_kernel, dim3(1), dim3(kDdThreads), 0, stream, nblk, N, blk, M);
  hipLaunchKernelGGL(dd_emit_kernel, dim3(nblk), dim3(kDdThreads), 0, stream, N, all, all, sk, sv, blk, M);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

int dedup_max_tables() { return kDedupMaxTables; }

static int dedup_build_map(const Dims& d, long long nnz, const int64_t* indices, const int64_t* tableidx, const DedupMap& M,
                           hipStream_t stream);
int dedup_build(const Dims& d, long long nnz, const int64_t* indices, const int64_t* tableidx, const DedupMap& M,
                hipStream_t stream) {
  const int rc = dedup_build_map(d, nnz, indices, tableidx, M, stream);
  if (rc) return rc;
  if (d.num_tables <= kDedupMaxTables && plan_groups_tables(d, nnz)) {
    ProfScope ps(TTX_PROF_PLAN, stream);
    hipLaunchKernelGGL(dedup_tstart_kernel, dim3((d.num_tables + 256) / 256), dim3(256), 0, stream, d.num_tables, M);
    TTX_HIP(hipGetLastError());
  }
  return TTX_OK;
}

static int dedup_build_map(const Dims& d, long long nnz, const int64_t* indices, const int64_t* tableidx, const DedupMap& M,
                           hipStream_t stream) {
  if (!dedup_supported(d, nnz)) TTX_FAIL(TTX_EUNSUPPORTED, "batch of %lld lookups / this key space is not deduplicated", nnz);
  if (nnz > kDedupMaxN || dedup_key_space(d) == 0) return dedup_build_large(d, nnz, indices, tableidx, M, stream);
  const unsigned long long all = dedup_key_space(d);
  const unsigned long long E = all / (unsigned long long)d.num_tables;
  int bits = 1;
  while (bits < 32 && (1ull << bits) < all) ++bits;
  const int passes = (bits + 7) / 8;
  const int N = (int)nnz;
  const size_t lds = (256 * kPlanWaves + 32 + 2 * (((size_t)N + 63) / 64 * 64)) * sizeof(int);
  const int per = ((N + kPlanWaves - 1) / kPlanWaves + kWave - 1) / kWave * kWave;
  const int nb = per / kWave;
  ProfScope ps(TTX_PROF_PLAN, stream);
#define TTX_DEDUP_LAUNCH(BPW)                                                                                 \
  do {                                                                                                        \
    const int rc_attr = allow_dynamic_lds((const void*)dedup_small_kernel<BPW>, 160 * 1024);                  \
    if (rc_attr) return rc_attr;                                                                              \
    hipLaunchKernelGGL(dedup_small_kernel<BPW>, dim3(1), dim3(kPlanThreads), lds, stream, N, E, d.num_tables, \
                       passes, indices, tableidx, M);                                                         \
  } while (0)
  if (nb <= 2) TTX_DEDUP_LAUNCH(2);
  else if (nb <= 4) TTX_DEDUP_LAUNCH(4);
  else if (nb <= 8) TTX_DEDUP_LAUNCH(8);
  else if (nb <= 12) TTX_DEDUP_LAUNCH(12);
  else TTX_DEDUP_LAUNCH(16);
#undef TTX_DEDUP_LAUNCH
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

// one launch for "offsets -> bag rows (+ frequency update) + plan" when the batch qualifies
bool prologue_fusable(const Dims& d, long long nnz, long long nb) {
  if (d.num_tables != 1 || !(nnz > 1024 && nnz <= kOneMaxN) || nb < 1 || nb > kProMaxBags) return false;
  for (int t = 0; t < d.T; ++t) if (d.S[t] > 256) return false;
  return true;
}

int prologue_launch(const Dims& d, int N, const int64_t* indices, const Prologue& pg, const Plan& P, hipStream_t stream,
                    const ProBatch* mb = nullptr, int nbatch = 1, const int* n_dev = nullptr) {
  for (int z = 0; z < nbatch && check_indices_on(); ++z) {  // (one table, tableidx == 0: the prologue's own contract)
    const int rc_chk = check_indices(d, N, n_dev, (mb && z > 0) ? mb->indices[z] : indices, nullptr, stream);
    if (rc_chk) return rc_chk;
  }
  ProfScope ps(TTX_PROF_PLAN, stream);
  hipLaunchKernelGGL(mb_single_kernel<true>,
                     dim3((N + kOneWaves * kOneUnit - 1) / (kOneWaves * kOneUnit) + TTX_PLAN_XWG, d.T, nbatch),
                     dim3(kOneThreads), 0, stream, d, N, n_dev, indices, nullptr, nullptr, P, pg,
                     mb ? *mb : ProBatch{});
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

}  // namespace ttx

extern "C" {

int ttx_lookup_prologue(const ttx_geom* g, int64_t nnz, const int64_t* colidx, int64_t nb, const int64_t* offsets,
                        int64_t H, int64_t* upd_hashtbl, int64_t* upd_cache_freq, int64_t* rowidx,
                        int64_t* tableidx, void* plan, size_t plan_bytes, ttx_stream_t stream) {
  return ttx_lookup_prologue_n(g, nnz, colidx, nb, offsets, H, upd_hashtbl, upd_cache_freq, rowidx, tableidx, plan, plan_bytes,
                               nullptr, stream);
}

// ... with the number of LIVE lookups on the device (nnz_dev, <= nnz): colidx holds nnz entries of which only the first
// *nnz_dev belong to a bag -- offsets[nb] == *nnz_dev -- and only those are planned: every kernel that later runs off this plan
// (forward, pooling, backward, reduce + apply) works on exactly the live lookups, whatever nnz their launches are sized for.
// What a table-sharded owner needs for RAGGED bags: the exchange buffers have a fixed capacity, the count stays on the device.
int ttx_lookup_prologue_n(const ttx_geom* g, int64_t nnz, const int64_t* colidx, int64_t nb, const int64_t* offsets,
                          int64_t H, int64_t* upd_hashtbl, int64_t* upd_cache_freq, int64_t* rowidx,
                          int64_t* tableidx, void* plan, size_t plan_bytes, const int32_t* nnz_dev, ttx_stream_t stream) {
  ttx::Dims d;
  int rc = ttx::make_dims(g, &d);
  if (rc != TTX_OK) return rc;
  if (nnz == 0) return TTX_OK;
  if (!colidx || !offsets || !rowidx || !tableidx) TTX_FAIL(TTX_EINVAL, "NULL input");
  if (nb <= 0 || nb % d.num_tables != 0) TTX_FAIL(TTX_EINVAL, "offsets must hold num_tables * B + 1 entries");
  if (!plan || plan_bytes < ttx::plan_bytes(d, nnz))
    TTX_FAIL(TTX_EWORKSPACE, "plan buffer too small: %zu < %zu", plan_bytes, ttx::plan_bytes(d, nnz));
  const bool upd = upd_hashtbl && upd_cache_freq;
  if (upd && (H <= 0 || H >= (1ll << 31))) TTX_FAIL(TTX_EINVAL, "hashtbl_size=%lld must be in (0, 2^31)", (long long)H);
  ttx::Plan P = ttx::carve_plan(d, nnz, plan);
  if (P.MC <= 0) TTX_FAIL(TTX_EUNSUPPORTED, "TT shape does not fit the LDS of any kernel variant (core-1 slice %d x %d floats)", d.k[0], d.n[0]);
  if (ttx::prologue_fusable(d, nnz, nb)) {
    ttx::Prologue pg{offsets, (int)nb, rowidx, tableidx, upd ? (int)H : 0, upd_hashtbl, upd_cache_freq};
    return ttx::prologue_launch(d, (int)nnz, colidx, pg, P, (hipStream_t)stream, nullptr, 1, nnz_dev);
  }
  if (nnz_dev && upd) TTX_FAIL(TTX_EUNSUPPORTED, "a device-side lookup count with a frequency table: not on this route");
  // general shape: the separate launches, same results
  int32_t ntt = 0, part = 0;
  rc = ttx_preprocess_indices_sync_fused(nnz, colidx, nb, offsets, d.num_tables, /*warmup=*/1, H, nullptr, nullptr,
                                         rowidx, tableidx, nullptr, nullptr, nullptr, &ntt, &part, upd_hashtbl,
                                         upd_cache_freq, nullptr, 0, stream);
  if (rc != TTX_OK) return rc;
  // (the bags are table-major by construction here: the plan may sort table groups on their own)
  // (nnz_dev: positions beyond the live count get the last bag's row from the offsets search -- nothing reads them)
  return ttx::plan_build(d, nnz, colidx, tableidx, rowidx, P, (hipStream_t)stream, nnz_dev, offsets,
                         (int)(nb / d.num_tables));
}

int ttx_lookup_prologue_multi(const ttx_geom* g, int32_t nbatch, int64_t nnz, const int64_t* const* colidx_host,
                              int64_t nb, const int64_t* const* offsets_host, int64_t H, int64_t* upd_hashtbl,
                              int64_t* upd_cache_freq, int64_t* rowidx, int64_t* tableidx, void* plans,
                              size_t plan_stride, ttx_stream_t stream) {
  ttx::Dims d;
  int rc = ttx::make_dims(g, &d);
  if (rc != TTX_OK) return rc;
  if (nbatch <= 0 || nnz == 0) return TTX_OK;
  if (!colidx_host || !offsets_host || !rowidx || !tableidx || !plans) TTX_FAIL(TTX_EINVAL, "NULL input");
  if (nb <= 0 || nb % d.num_tables != 0) TTX_FAIL(TTX_EINVAL, "offsets must hold num_tables * B + 1 entries");
  const size_t pb = ttx::plan_bytes(d, nnz);
  if (plan_stride < pb || plan_stride % 256 != 0)
    TTX_FAIL(TTX_EWORKSPACE, "plan stride %zu: need a multiple of 256 of at least %zu bytes", plan_stride, pb);
  const bool upd = upd_hashtbl && upd_cache_freq;
  if (upd && (H <= 0 || H >= (1ll << 31))) TTX_FAIL(TTX_EINVAL, "hashtbl_size=%lld must be in (0, 2^31)", (long long)H);
  for (int z = 0; z < nbatch; ++z)
    if (!colidx_host[z] || !offsets_host[z]) TTX_FAIL(TTX_EINVAL, "batch %d: NULL indices / offsets", z);
  if (!ttx::prologue_fusable(d, nnz, nb)) {  // general shape: batch after batch, same results
    for (int z = 0; z < nbatch; ++z) {
      rc = ttx_lookup_prologue(g, nnz, colidx_host[z], nb, offsets_host[z], H, upd_hashtbl, upd_cache_freq,
                               rowidx + (size_t)z * nnz, tableidx + (size_t)z * nnz, (char*)plans + (size_t)z * plan_stride,
                               plan_stride, stream);
      if (rc != TTX_OK) return rc;
    }
    return TTX_OK;
  }
  for (int z0 = 0; z0 < nbatch; z0 += ttx::kMaxMulti) {  // kMaxMulti batches per launch
    const int nz = nbatch - z0 < ttx::kMaxMulti ? nbatch - z0 : ttx::kMaxMulti;
    ttx::ProBatch mb{};
    for (int z = 0; z < nz; ++z) { mb.indices[z] = colidx_host[z0 + z]; mb.offsets[z] = offsets_host[z0 + z]; }
    mb.out_stride = nnz;
    mb.plan_stride = (long long)plan_stride;
    ttx::Plan P = ttx::carve_plan(d, nnz, (char*)plans + (size_t)z0 * plan_stride);
    ttx::Prologue pg{offsets_host[z0], (int)nb, rowidx + (size_t)z0 * nnz, tableidx + (size_t)z0 * nnz, upd ? (int)H : 0,
                     upd_hashtbl, upd_cache_freq};
    rc = ttx::prologue_launch(d, (int)nnz, colidx_host[z0], pg, P, (hipStream_t)stream, &mb, nz);
    if (rc != TTX_OK) return rc;
  }
  return TTX_OK;
}

size_t ttx_plan_bytes(const ttx_geom* g, int64_t nnz) {
  ttx::Dims d;
  if (ttx::make_dims(g, &d) != TTX_OK || nnz < 0) return 0;
  return ttx::plan_bytes(d, nnz);
}

int ttx_plan_build(const ttx_geom* g, int64_t nnz, const int64_t* indices,
                   const int64_t* tableidx, const int64_t* rowidx, void* plan, size_t plan_bytes,
                   ttx_stream_t stream) {
  return ttx_plan_build_n(g, nnz, nullptr, indices, tableidx, rowidx, plan, plan_bytes, stream);
}

int ttx_plan_build_n(const ttx_geom* g, int64_t nnz, const int32_t* nnz_dev, const int64_t* indices,
                     const int64_t* tableidx, const int64_t* rowidx, void* plan, size_t plan_bytes,
                     ttx_stream_t stream) {
  ttx::Dims d;
  int rc = ttx::make_dims(g, &d);
  if (rc != TTX_OK) return rc;
  if (!plan || plan_bytes < ttx::plan_bytes(d, nnz))
    TTX_FAIL(TTX_EWORKSPACE, "plan buffer too small: %zu < %zu", plan_bytes, ttx::plan_bytes(d, nnz));
  if (nnz > 0 && !indices) TTX_FAIL(TTX_EINVAL, "indices is NULL");
  ttx::Plan P = ttx::carve_plan(d, nnz, plan);
  return ttx::plan_build(d, nnz, indices, tableidx, rowidx, P, (hipStream_t)stream, nnz_dev);
}

}  // extern "C"
