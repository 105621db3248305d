// ttx_plan.hip -- lookup plan: index decode + stable radix sorts by core slice.
//
// Replaces the reference's per-chunk set-up kernels
// (init_batch_gemm_{forward,backward}_*T_kernel, tt_embeddings_cuda.cu:79-360,
// :754-918) and the need for float atomics in update_d_tt_cores_kernel
// (:362-377): after this kernel every core slice knows, in index order, the
// lookups that touch it, so gradients are reduced by one owner per slice.
//
// One launch, T work-groups of 1024 threads (16 waves); work-group t plans core t:
//   1. decode idx -> i_t exactly as the reference (idx / L_t % p_t, cu:795-799;
//      the int64 divisions are done as a double-reciprocal multiply + fix-up
//      when prod(p) < 2^31) and form the slice id  sid[t][n] = table*p_t + i_t;
//   2. a stable LSD radix sort (8-bit digits) of the lookups by sid[t].  Ranking
//      inside a wave uses 9 ballots per 64 keys (wave_match8); per-wave digit
//      counters live in LDS and are combined by one block-wide exclusive scan
//      per pass -- no atomics, deterministic;
//   3. slice offsets by run-head detection on the sorted keys;
//   4. (pivot core 1 only) the chunk work-list and one flat record per sorted
//      lookup {n, sid_0, sid_2, sid_3}, so a contraction work-group reaches its
//      operands with two dependent loads instead of a five-deep pointer chase.
// The plan is HBM-resident integer work: ~ (8 + 8*passes) bytes per lookup per
// core; at the benchmark shape (nnz 10240, 200/220/250 slices) every sort is a
// single 8-bit pass.
#include "ttx_internal.h"

namespace ttx {

constexpr int kPlanThreads = 1024;
constexpr int kPlanWaves = kPlanThreads / kWave;

int max_chunks(const Dims& d, long long nnz, int MC) {
  long long a = nnz < d.S[1] ? nnz : d.S[1];
  long long v = a + nnz / MC + 1;
  return (int)v;
}

static size_t r64(size_t k) { return (k + 63) / 64 * 64; }

static size_t plan_ints(const Dims& d, long long nnz, int MC) {
  size_t n = r64(64);                                               // hdr
  for (int t = 0; t < d.T; ++t) n += r64(nnz) * 5 + r64((size_t)d.S[t] + 1);  // sid, perm, 3 scratch, off
  n += r64((size_t)d.S[1] + 1);                                     // chunk_off
  n += r64((size_t)max_chunks(d, nnz, MC) * 4);                     // chunk_rec
  n += r64((size_t)nnz * 4);                                        // lrec
  return n;
}

size_t plan_bytes(const Dims& d, long long nnz) {
  return align_up(plan_ints(d, nnz, choose_chunk(d)) * sizeof(int));
}

Plan carve_plan(const Dims& d, long long nnz, void* base) {
  Plan P;
  memset(&P, 0, sizeof(P));
  P.MC = choose_chunk(d);
  P.max_chunks = max_chunks(d, nnz, P.MC);
  int* cur = (int*)base;
  auto take = [&](size_t k) { int* r = cur; cur += r64(k); return r; };
  P.hdr = take(64);
  P.chunk_rec = (int4*)take((size_t)P.max_chunks * 4);
  P.lrec = (int4*)take((size_t)nnz * 4);
  for (int t = 0; t < d.T; ++t) {
    P.sid[t] = take(nnz);
    P.perm[t] = take(nnz);
    P.off[t] = take((size_t)d.S[t] + 1);
    for (int i = 0; i < 3; ++i) P.scratch[t][i] = take(nnz);
  }
  P.chunk_off = take((size_t)d.S[1] + 1);
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

// floor(n / dv) for 0 <= n < 2^31 via one fp64 multiply and a fix-up
__device__ __forceinline__ unsigned div_small(unsigned n, unsigned dv, double rcp) {
  unsigned q = (unsigned)((double)n * rcp);
  const long long r = (long long)n - (long long)q * dv;
  if (r < 0) q -= 1;
  else if (r >= (long long)dv) q += 1;
  return q;
}

// i_t of lookup idx (reference decode: i_t = idx / L_t % p_t; out-of-range
// indices, which the reference reads out of bounds with, are clamped)
__device__ __forceinline__ int decode_core(const Dims& d, int t, long long idx, bool small,
                                           double rcpL, double rcpP) {
  if (idx < 0) idx = 0;
  long long a;
  if (small && idx < (1ll << 31)) {
    const unsigned q = div_small((unsigned)idx, (unsigned)d.L[t], rcpL);
    if (t == 0) a = q;
    else a = q - div_small(q, (unsigned)d.p[t], rcpP) * (unsigned)d.p[t];
  } else {
    a = idx / d.L[t];
    if (t > 0) a = a % d.p[t];
  }
  if (a >= d.p[t]) a = d.p[t] - 1;
  return (int)a;
}

__global__ __launch_bounds__(kPlanThreads) void plan_kernel(
    Dims d, int N, const int64_t* __restrict__ indices, const int64_t* __restrict__ tableidx, Plan P) {
  __shared__ int hist[256 * kPlanWaves];  // [digit][wave]
  __shared__ int wtot[kPlanWaves + 1];
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int w = tid / kWave;
  const int t = blockIdx.x;  // the core this work-group plans
  const bool small = d.L[0] * (long long)d.p[0] < (1ll << 31);
  int* key = P.sid[t];

  // ---- 1. decode ----------------------------------------------------------
  {
    const double rcpL = 1.0 / (double)d.L[t], rcpP = 1.0 / (double)d.p[t];
    for (int n = tid; n < N; n += kPlanThreads) {
      const long long tb = tableidx ? tableidx[n] : 0;
      key[n] = (int)(tb * d.p[t]) + decode_core(d, t, indices[n], small, rcpL, rcpP);
    }
  }
  __syncthreads();

  // per-wave contiguous range of the current order
  const int per = ((N + kPlanWaves - 1) / kPlanWaves + kWave - 1) / kWave * kWave;
  const int wbeg = w * per;
  const int wend = min(N, wbeg + per);

  // ---- 2. stable LSD radix sort by sid[t] ----------------------------------
  {
    int bits = 32 - __clz(max(d.S[t] - 1, 1));
    int passes = (bits + 7) / 8;
    if (passes < 1) passes = 1;
    int* rk = P.scratch[t][0];
    for (int ps = 0; ps < passes; ++ps) {
      const int shift = ps * 8;
      const int* src = (ps == 0) ? nullptr : ((ps & 1) ? P.scratch[t][1] : P.scratch[t][2]);
      int* dst = (ps == passes - 1) ? P.perm[t] : ((ps & 1) ? P.scratch[t][2] : P.scratch[t][1]);
      for (int e = tid; e < 256 * kPlanWaves; e += kPlanThreads) hist[e] = 0;
      __syncthreads();
      // count + rank inside the wave's range
      for (int base = wbeg; base < wend; base += kWave) {
        const int i = base + lane;
        const bool valid = i < wend;
        int val = 0;
        unsigned dg = 0;
        if (valid) {
          val = src ? src[i] : i;
          dg = ((unsigned)key[val] >> shift) & 255u;
        }
        const unsigned long long peers = wave_match8(dg, valid);
        if (valid) {
          const int before = hist[dg * kPlanWaves + w];
          rk[i] = before + __popcll(peers & lanemask_lt());
          // the lowest peer publishes the new count (LDS ops of one wave are
          // executed in program order: every peer read `before` already)
          if ((peers & lanemask_lt()) == 0) hist[dg * kPlanWaves + w] = before + __popcll(peers);
        }
      }
      __syncthreads();
      // exclusive scan over (digit major, wave minor): 4 entries per thread
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
      // scatter
      for (int base = wbeg; base < wend; base += kWave) {
        const int i = base + lane;
        if (i < wend) {
          const int val = src ? src[i] : i;
          const unsigned dg = ((unsigned)key[val] >> shift) & 255u;
          dst[hist[dg * kPlanWaves + w] + rk[i]] = val;
        }
      }
      __syncthreads();
    }
  }
  // ---- 3. slice offsets by run-head detection -------------------------------
  const int* pm = P.perm[t];
  {
    int* off = P.off[t];
    const int S = d.S[t];
    for (int i = tid; i <= N; i += kPlanThreads) {
      const int kprev = (i == 0) ? -1 : key[pm[i - 1]];
      const int kcur = (i == N) ? S : key[pm[i]];
      for (int s = kprev + 1; s <= kcur; ++s) off[s] = i;
    }
  }
  if (t != 1) return;
  __syncthreads();

  // ---- 4. pivot core: chunk work-list + flat per-lookup records ------------
  {
    const int S1 = d.S[1];
    const int MC = P.MC;
    const int* off = P.off[1];
    int carry = 0;
    for (int s0 = 0; s0 < S1; s0 += kPlanThreads) {
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
        for (int j = 0; j < nch; ++j)
          P.chunk_rec[ex + j] = make_int4(s, beg + j * MC, min(MC, cnt - j * MC), 0);
      }
      carry += total;
    }
    // unused tail of the work list: len = 0 -> the contraction work-group exits
    for (int c = carry + tid; c < P.max_chunks; c += kPlanThreads) P.chunk_rec[c] = make_int4(0, 0, 0, 0);
    if (tid == 0) {
      P.chunk_off[S1] = carry;
      P.hdr[0] = carry;
      P.hdr[1] = MC;
      P.hdr[2] = N;
    }
    double rcpL[TTX_MAX_CORES], rcpP[TTX_MAX_CORES];
#pragma unroll
    for (int u = 0; u < TTX_MAX_CORES; ++u) {
      rcpL[u] = u < d.T ? 1.0 / (double)d.L[u] : 1.0;
      rcpP[u] = u < d.T ? 1.0 / (double)d.p[u] : 1.0;
    }
    for (int i = tid; i < N; i += kPlanThreads) {
      const int n = pm[i];
      const long long idx = indices[n];
      const int tb = tableidx ? (int)tableidx[n] : 0;
      int4 r;
      r.x = n;
      r.y = tb * d.p[0] + decode_core(d, 0, idx, small, rcpL[0], rcpP[0]);
      r.z = d.T > 2 ? tb * d.p[2] + decode_core(d, 2, idx, small, rcpL[2], rcpP[2]) : 0;
      r.w = d.T > 3 ? tb * d.p[3] + decode_core(d, 3, idx, small, rcpL[3], rcpP[3]) : 0;
      P.lrec[i] = r;
    }
  }
}

int plan_build(const Dims& d, long long nnz, const int64_t* indices,
               const int64_t* tableidx, const Plan& P, hipStream_t stream) {
  if (nnz < 0 || nnz >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "nnz=%lld out of range", nnz);
  ProfScope ps(TTX_PROF_PLAN, stream);
  hipLaunchKernelGGL(plan_kernel, dim3(d.T), dim3(kPlanThreads), 0, stream, d, (int)nnz,
                     indices, tableidx, P);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

}  // namespace ttx

extern "C" {

size_t ttx_plan_bytes(const ttx_geom* g, int64_t nnz) {
  ttx::Dims d;
  if (ttx::make_dims(g, &d) != TTX_OK || nnz < 0) return 0;
  return ttx::plan_bytes(d, nnz);
}

int ttx_plan_build(const ttx_geom* g, int64_t nnz, const int64_t* indices,
                   const int64_t* tableidx, void* plan, size_t plan_bytes,
                   ttx_stream_t stream) {
  ttx::Dims d;
  int rc = ttx::make_dims(g, &d);
  if (rc != TTX_OK) return rc;
  if (!plan || plan_bytes < ttx::plan_bytes(d, nnz))
    TTX_FAIL(TTX_EWORKSPACE, "plan buffer too small: %zu < %zu", plan_bytes, ttx::plan_bytes(d, nnz));
  if (nnz > 0 && !indices) TTX_FAIL(TTX_EINVAL, "indices is NULL");
  ttx::Plan P = ttx::carve_plan(d, nnz, plan);
  return ttx::plan_build(d, nnz, indices, tableidx, P, (hipStream_t)stream);
}

}  // extern "C"
