// ttx_tt.hip -- TT-core contraction forward / backward for gfx950 (CDNA4).
//
// Design: lookups are grouped by the slice of the PIVOT core
// (core 1, the big r1 x q1 x r2 slice) they touch.  One work-group (4 waves)
// owns a chunk of <= MC lookups of one pivot slice:
//   forward   X0[M x N1] = A[M x r1] * B1[r1 x N1]     M = MC*q0, N1 = q1*r2
//             (B1 staged ONCE in LDS, A = the chunk's core-0 slices stacked) on
//             v_mfma_f32_16x16x4_f32 -- exact fp32, bit-identical to an fmaf
//             chain -- then the remaining T-2 stages per lookup out of LDS,
//             rows -> HBM, bags pooled by a second tiny kernel in index order
//             (the order of the reference's reduce_output_kernel,
//             tt_embeddings_cuda.cu:920-962);
//   backward  recompute X0, per-lookup tail (grad of the last cores), then two
//             more chunk GEMMs on MFMA:  dB1 = A^T * dX0  and  dA = dX0 * B1^T.
// Gradients never use atomics: every producer writes a private partial (per
// lookup for the thin cores, per chunk for the pivot) and ONE owner per core
// slice sums them in index order and applies DENSE / SGD / Adagrad
// (reduce_apply_kernel) -- deterministic, and the optimizer touches only
// slices that were looked up.
//
// LDS layout: every operand is row-major with a row stride == 2 (mod 32) floats.
// With that single choice all three GEMMs read their MFMA fragments with
// ds_read_b32 conflict-free (or 2-way on the operand that is re-used from
// registers), by choosing which 4 K-indices a k-step covers:
//   X0  = A  * B1   k-step {k, k+8, k+16, k+24}  (B rows 8 apart: 8*ld == 16 mod 32)
//   dB1 = A^T* dX0  k-step {m, m+8, m+16, m+24}  (both operands row-strided)
//   dA  = dX0* B1^T k-step {c, c+1, c+2, c+3}    (both operands column-adjacent)
#include "ttx_tt_common.h"

namespace ttx {

#include "ttx_tt_generic.inc"

// out[table,row,:] += sum of the run's rows, in index order (run = consecutive
// lookups with equal (rowidx, tableidx); reference reduce_output_kernel
// cu:920-962).  One 32-lane group per lookup; only run heads work.  The run
// length is found with one ballot per 32 candidates.
__global__ __launch_bounds__(kThreads) void pool_kernel(int Nmax, const int* __restrict__ hdr, int B, int D,
                                                       const int64_t* __restrict__ rowidx,
                                                       const int64_t* __restrict__ tableidx,
                                                       const float* __restrict__ rows,
                                                       const float* __restrict__ psw, float* __restrict__ out) {
  const int n = blockIdx.x * (kThreads / 32) + threadIdx.x / 32;
  const int l = threadIdx.x & 31;
  const int N = min(Nmax, hdr[2]);  // the plan knows how many lookups are live (device-side counts)
  if (n >= N) return;
  const int64_t r = rowidx[n], tb = tableidx[n];
  if (n > 0 && rowidx[n - 1] == r && tableidx[n - 1] == tb) return;
  // half-wave cooperative scan for the end of the run
  const unsigned long long half = (threadIdx.x & 32) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
  int sl = 1;
  for (;;) {
    const int c = n + sl + l;
    const bool same = c < N && rowidx[c] == r && tableidx[c] == tb;
    unsigned long long m = (__ballot(!same) & half) >> (threadIdx.x & 32);
    if (m) { sl += __builtin_ctzll(m); break; }
    sl += 32;
  }
  float* o = out + ((size_t)tb * B + r) * D;
  const float* src = rows + (size_t)n * D;
  if (psw) {  // weighted sum (per_sample_weights), same order
    for (int e = l; e < D; e += 32) {
      float acc = o[e];
      for (int j = 0; j < sl; ++j) acc = fmaf(psw[n + j], src[(size_t)j * D + e], acc);
      o[e] = acc;
    }
    return;
  }
  for (int e = l; e < D; e += 32) {
    float acc = o[e];
    int j = 0;
    for (; j + 8 <= sl; j += 8) {  // eight rows in flight, added in index order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(j + u) * D + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; j + 4 <= sl; j += 4) {
      const float v0 = src[(size_t)j * D + e], v1 = src[(size_t)(j + 1) * D + e];
      const float v2 = src[(size_t)(j + 2) * D + e], v3 = src[(size_t)(j + 3) * D + e];
      acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; j < sl; ++j) acc += src[(size_t)j * D + e];
    o[e] = acc;
  }
}

// the same for D % 4 == 0, at memory speed: a wave owns 64 consecutive lookups, finds the run heads
// among them with one ballot, and its four 16-lane groups take those heads round-robin (16-byte
// loads: a 64-float row is one load per lane; a run is summed in index order by ONE group, eight
// rows in flight).  pool_kernel keeps one half-wave in twenty busy at 20 lookups per bag.
__global__ __launch_bounds__(kThreads) void pool4_kernel(int Nmax, const int* __restrict__ hdr, int B, int D4,
                                                        const int64_t* __restrict__ rowidx,
                                                        const int64_t* __restrict__ tableidx,
                                                        const float4* __restrict__ rows,
                                                        const float* __restrict__ psw, float4* __restrict__ out) {
  __shared__ unsigned char hpos[kThreads / kWave][kWave];
  const int w = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  const int N = min(Nmax, hdr[2]);  // the plan knows how many lookups are live (device-side counts)
  const int n0 = (blockIdx.x * (kThreads / kWave) + w) * kWave;
  if (n0 >= N) return;  // (wave-uniform)
  const int n = n0 + lane;
  bool head = false;
  if (n < N) head = n == 0 || rowidx[n - 1] != rowidx[n] || tableidx[n - 1] != tableidx[n];
  const unsigned long long heads = __ballot(head);
  if (head) hpos[w][__popcll(heads & ((1ull << lane) - 1ull))] = (unsigned char)lane;
  const int nheads = __popcll(heads);
  const int g = lane >> 4, l = lane & 15, sh = lane & 48;
  for (int k = g; k < nheads; k += 4) {
    const int pos = hpos[w][k];
    const int hn = n0 + pos;
    const int64_t r = rowidx[hn], tb = tableidx[hn];
    int sl;
    if (k + 1 < nheads) {
      sl = hpos[w][k + 1] - pos;
    } else {  // the span's last run may go on behind it: 16 candidates per ballot (one group gets here)
      sl = min(N, n0 + kWave) - hn;
      if (hn + sl < N)
        for (;;) {
          const int c = hn + sl + l;
          const bool same = c < N && rowidx[c] == r && tableidx[c] == tb;
          const unsigned m = (unsigned)(__ballot(!same) >> sh) & 0xffffu;
          if (m) { sl += __builtin_ctz(m); break; }
          sl += 16;
        }
    }
    float4* o = out + ((size_t)tb * B + r) * D4;
    const float4* src = rows + (size_t)hn * D4;
    for (int e = l; e < D4; e += 16) {
      float4 acc = o[e];
      if (psw) {  // weighted sum (per_sample_weights), same order
        for (int j = 0; j < sl; ++j) {
          const float wj = psw[hn + j];
          const float4 v = src[(size_t)j * D4 + e];
          acc.x = fmaf(wj, v.x, acc.x); acc.y = fmaf(wj, v.y, acc.y); acc.z = fmaf(wj, v.z, acc.z); acc.w = fmaf(wj, v.w, acc.w);
        }
      } else {
        int j = 0;
        for (; j + 8 <= sl; j += 8) {
          float4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(j + u) * D4 + e];
#pragma unroll
          for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; j + 4 <= sl; j += 4) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = src[(size_t)(j + u) * D4 + e];
#pragma unroll
          for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; j < sl; ++j) {
          const float4 v = src[(size_t)j * D4 + e];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
      o[e] = acc;
    }
  }
}

// ... and for small batches (<= kPoolSpanMin lookups), where latency counts and not bandwidth: one
// 16-lane group per lookup, only the run heads work -- every bag in parallel.  (Short bags make the
// wave-span kernel walk up to 16 heads per group one after the other: 12 us for 1200 lookups in 512
// bags.)  Same sums in the same order.
constexpr int kPoolSpanMin = 65536;
// (Round 6: the chain of dependent trips to memory is what this kernel costs -- live count -> this lookup's bag -> the run
//  scan -> the rows, 4.9 us for 2.6 MB at cfg2.  Everything whose ADDRESS does not depend on a loaded value is requested in one
//  round: the live count, this lookup's bag and its predecessor's, the 32 candidates of the first two scan rounds (positions
//  clamped to the arrays' last entry, compared under the real bound); then, the run length known, up to 32 rows of the run are in
//  flight at once.  Longer runs go on as before.  Same sums in the same order.)
constexpr int kPoolAhead = 24;  // rows in flight per round
// N 64-bit (one 32-bit) values requested back to back and awaited together.  Written in assembly because the compiler sinks a load
// whose value is first read behind a branch to behind that branch (a lookup that is not a run head leaves early): the requests
// for this lookup's bag, its predecessor's and the scan candidates -- all at known addresses -- became three trips to memory.
__device__ __forceinline__ void pool_ld_round(const int64_t* const (&p)[8], int64_t (&v)[8]) {
  asm volatile(
      "global_load_dwordx2 %0, %8, off\n\t"
      "global_load_dwordx2 %1, %9, off\n\t"
      "global_load_dwordx2 %2, %10, off\n\t"
      "global_load_dwordx2 %3, %11, off\n\t"
      "global_load_dwordx2 %4, %12, off\n\t"
      "global_load_dwordx2 %5, %13, off\n\t"
      "global_load_dwordx2 %6, %14, off\n\t"
      "global_load_dwordx2 %7, %15, off\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
      : "memory");
}
__device__ __forceinline__ void pool_ld_round(const int64_t* const (&p)[4], const int32_t* q, int64_t (&v)[4], int32_t& w) {
  asm volatile(
      "global_load_dwordx2 %0, %5, off\n\t"
      "global_load_dwordx2 %1, %6, off\n\t"
      "global_load_dwordx2 %2, %7, off\n\t"
      "global_load_dwordx2 %3, %8, off\n\t"
      "global_load_dword %4, %9, off\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(w)
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(q)
      : "memory");
}
// (rows through a buffer descriptor over the whole array: one 32-bit offset register per load instead of a 64-bit address, and a
//  row behind the run's end is an offset behind the buffer's -- the hardware returns zeros, no branch around the load)
typedef unsigned int pool_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pool_ld4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const pool_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pool_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__global__ __launch_bounds__(kThreads, 3) void pool4_small_kernel(const int* __restrict__ hdr, const int64_t* __restrict__ rowidx,
                                                                 const int64_t* __restrict__ tableidx,
                                                                 const float4* __restrict__ rows, float4* __restrict__ out,
                                                                 int Nmax, int B, int D4, const float* __restrict__ psw) {
  // (argument order: the first 14 dwords arrive in SGPRs with the wave -- everything but the per_sample_weights)
  const int n = blockIdx.x * (kThreads / 16) + threadIdx.x / 16;
  const int l = threadIdx.x & 15;
  if (n >= Nmax) return;  // (the arrays hold Nmax entries: every position below is safe to request whatever the live count)
  const int last = Nmax - 1;
  const int c1 = n + 1 + l, c2 = n + 17 + l;
  const int nlive = hdr[2];
  const int np = n > 0 ? n - 1 : 0, k1 = min(c1, last), k2 = min(c2, last);
  const int64_t* const pp[8] = {rowidx + n, tableidx + n, rowidx + np, tableidx + np, rowidx + k1, tableidx + k1, rowidx + k2, tableidx + k2};
  int64_t vv[8];
  pool_ld_round(pp, vv);
  const int64_t r = vv[0], tb = vv[1], rp = vv[2], tp = vv[3], r1 = vv[4], t1 = vv[5], r2 = vv[6], t2 = vv[7];
  const int N = min(Nmax, nlive);
  if (n >= N) return;
  if (n > 0 && rp == r && tp == tb) return;
  const int sh = threadIdx.x & 48;  // this group's 16 bits of the wave ballot
  int sl = 1;
  {
    const unsigned m1 = (unsigned)(__ballot(!(c1 < N && r1 == r && t1 == tb)) >> sh) & 0xffffu;
    const unsigned m2 = (unsigned)(__ballot(!(c2 < N && r2 == r && t2 == tb)) >> sh) & 0xffffu;
    if (m1) sl += __builtin_ctz(m1);
    else if (m2) sl += 16 + __builtin_ctz(m2);
    else {
      sl += 32;
      for (;;) {
        const int c = n + sl + l;
        const bool same = c < N && rowidx[c] == r && tableidx[c] == tb;
        const unsigned m = (unsigned)(__ballot(!same) >> sh) & 0xffffu;
        if (m) { sl += __builtin_ctz(m); break; }
        sl += 16;
      }
    }
  }
  float4* o = out + ((size_t)tb * B + r) * D4;
  const float4* src = rows + (size_t)n * D4;
  const __amdgpu_buffer_rsrc_t rrows = pool_rsrc(rows, (unsigned)Nmax * (unsigned)D4 * 16u);  // (< 2^32: the launcher's condition)
  for (int e = l; e < D4; e += 16) {
    float4 acc = o[e];
    if (psw) {
      for (int j = 0; j < sl; ++j) {
        const float wj = psw[n + j];
        const float4 v = src[(size_t)j * D4 + e];
        acc.x = fmaf(wj, v.x, acc.x); acc.y = fmaf(wj, v.y, acc.y); acc.z = fmaf(wj, v.z, acc.z); acc.w = fmaf(wj, v.w, acc.w);
      }
    } else {
      for (int j0 = 0; j0 < sl; j0 += kPoolAhead) {
        float4 v[kPoolAhead];
#pragma unroll
        for (int u = 0; u < kPoolAhead; ++u)
          v[u] = pool_ld4(rrows, j0 + u < sl ? ((unsigned)(n + j0 + u) * (unsigned)D4 + (unsigned)e) * 16u : 0xfffffff0u);
#pragma unroll
        for (int u = 0; u < kPoolAhead; ++u)
          if (j0 + u < sl) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
    }
    o[e] = acc;
  }
}

// Cache live (one table): the bag sums of the contraction's rows AND of the cache's rows in ONE launch.  The batch is
// partitioned -- misses at [0, n_tt) in index order, hits behind them (cache locations in loc[]) -- and a bag may have a run in
// both parts; the reference adds the cache rows in a second kernel (cache_forward, cu:1498-1572), this path used to do the same
// (pool4_small_kernel, then cache_forward4_kernel reading the output back).  Here every run head of either part adds ITS sum to
// the zeroed output row with fp32 atomics: a row receives at most two terms, 0 + A + B is the same number in either order, so the
// result does not depend on the order of arrival.  One launch less per step and the gather overlaps the pooling (cfg3).
// (Round 6, as pool4_small_kernel: one round of requests for everything whose address is known up front -- here also the cache
//  locations of the first 16 lookups of the run, handed round the 16-lane group by lane broadcasts -- 8.3 us at cfg3 before.)
constexpr int kPoolAheadC = 16;
__global__ __launch_bounds__(kThreads) void pool4_small_cached_kernel(const int* __restrict__ hdr, const int64_t* __restrict__ rowidx,
                                                                     const float4* __restrict__ rows,
                                                                     const int32_t* __restrict__ loc,
                                                                     const float4* __restrict__ cw, float* __restrict__ out,
                                                                     int nnz, int D4) {
  const int n = blockIdx.x * (kThreads / 16) + threadIdx.x / 16;
  const int l = threadIdx.x & 15;
  if (n >= nnz) return;
  const int last = nnz - 1;
  const int c1 = n + 1 + l, c2 = n + 17 + l;
  const int nlive = hdr[2];
  const int64_t* const pp[4] = {rowidx + n, rowidx + (n > 0 ? n - 1 : 0), rowidx + min(c1, last), rowidx + min(c2, last)};
  int64_t vv[4];
  int32_t la;  // cache location of lookup n + l
  pool_ld_round(pp, loc + min(n + l, last), vv, la);
  const int64_t r = vv[0], rp = vv[1], r1 = vv[2], r2 = vv[3];
  const int ntt = min(nnz, nlive);           // (the plan was built over the misses: its live count is the split point)
  const bool hit = n >= ntt;
  const int lo = hit ? ntt : 0, hi = hit ? nnz : ntt;  // this lookup's part
  if (n > lo && rp == r) return;             // not a run head
  const int sh = threadIdx.x & 48;           // this group's 16 bits of the wave ballot
  int sl = 1;
  {
    const unsigned m1 = (unsigned)(__ballot(!(c1 < hi && r1 == r)) >> sh) & 0xffffu;
    const unsigned m2 = (unsigned)(__ballot(!(c2 < hi && r2 == r)) >> sh) & 0xffffu;
    if (m1) sl += __builtin_ctz(m1);
    else if (m2) sl += 16 + __builtin_ctz(m2);
    else {
      sl += 32;
      for (;;) {
        const int c = n + sl + l;
        const bool same = c < hi && rowidx[c] == r;
        const unsigned m = (unsigned)(__ballot(!same) >> sh) & 0xffffu;
        if (m) { sl += __builtin_ctz(m); break; }
        sl += 16;
      }
    }
  }
  // the first 16 cache locations of the run, handed round the group while all of its 16 lanes are here
  int lj[kPoolAheadC];
#pragma unroll
  for (int u = 0; u < kPoolAheadC; ++u) lj[u] = __shfl(la, sh + u, kWave);
  float* o = out + (size_t)r * D4 * 4;
  for (int e = l; e < D4; e += 16) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = 0; j0 < sl; j0 += kPoolAheadC) {  // sixteen rows in flight, added in index order
      float4 v[kPoolAheadC];
#pragma unroll
      for (int u = 0; u < kPoolAheadC; ++u) {
        const int j = j0 + u;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < sl) {
          if (hit) {
            const int lc = j0 == 0 ? lj[u] : loc[n + j];
            v[u] = cw[(size_t)lc * D4 + e];
          } else {
            v[u] = rows[(size_t)(n + j) * D4 + e];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kPoolAheadC; ++u)
        if (j0 + u < sl) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    unsafeAtomicAdd(o + 4 * e, acc.x);
    unsafeAtomicAdd(o + 4 * e + 1, acc.y);
    unsafeAtomicAdd(o + 4 * e + 2, acc.z);
    unsafeAtomicAdd(o + 4 * e + 3, acc.w);
  }
}

// ---- duplicate lookups (DedupMap): pooling through uid[], and the bag gradients of a pair's occurrences ----
// Bag sums when the contraction ran once per DISTINCT (table, index) pair: row of lookup n = rows[uid[n]].  Same
// runs, same order of addition as pool4_small_kernel (index order), so the output is bit-identical to the plain
// path's.  One 16-lane group per lookup, only the run heads work.  V = float4 (D % 4 == 0) or float.
template <typename V>
__device__ __forceinline__ void vfma(V& acc, float w, const V& v);
template <>
__device__ __forceinline__ void vfma<float4>(float4& acc, float w, const float4& v) {
  acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}
template <>
__device__ __forceinline__ void vfma<float>(float& acc, float w, const float& v) { acc = fmaf(w, v, acc); }
__device__ __forceinline__ void vadd(float4& acc, const float4& v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
__device__ __forceinline__ void vadd(float& acc, const float& v) { acc += v; }
__device__ __forceinline__ void vzero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vzero(float& a) { a = 0.f; }

template <typename V>
__global__ __launch_bounds__(kThreads) void pool_gather_kernel(int N, int B, int DV, const int64_t* __restrict__ rowidx,
                                                              const int64_t* __restrict__ tableidx,
                                                              const int* __restrict__ uid, const V* __restrict__ rows,
                                                              const float* __restrict__ psw, V* __restrict__ out) {
  const int n = blockIdx.x * (kThreads / 16) + threadIdx.x / 16;
  const int l = threadIdx.x & 15;
  if (n >= N) return;
  const int64_t r = rowidx[n], tb = tableidx[n];
  if (n > 0 && rowidx[n - 1] == r && tableidx[n - 1] == tb) return;
  const int sh = threadIdx.x & 48;  // this group's 16 bits of the wave ballot
  int sl = 1;
  for (;;) {
    const int c = n + sl + l;
    const bool same = c < N && rowidx[c] == r && tableidx[c] == tb;
    const unsigned m = (unsigned)(__ballot(!same) >> sh) & 0xffffu;
    if (m) { sl += __builtin_ctz(m); break; }
    sl += 16;
  }
  V* o = out + ((size_t)tb * B + r) * DV;
  for (int eb = 0; eb < DV; eb += 16) {
    const int e = eb + l;
    const bool ev = e < DV;
    V acc;
    vzero(acc);
    if (ev) acc = o[e];
    for (int base = 0; base < sl; base += 16) {  // the lanes fetch 16 lookups' map entries at once, then 16 rows in flight
      const int j = base + l;
      const int u = j < sl ? uid[n + j] : 0;
      const float w = (psw && j < sl) ? psw[n + j] : 1.f;
      const int cnt = min(16, sl - base);
      V v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int uq = __shfl(u, q, 16);
        if (q < cnt && ev) v[q] = rows[(size_t)uq * DV + e];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {  // added in index order
        const float wq = __shfl(w, q, 16);
        if (q < cnt && ev) {
          if (psw) vfma(acc, wq, v[q]);
          else vadd(acc, v[q]);
        }
      }
    }
    if (ev) o[e] = acc;
  }
}

// the same for large batches (D % 4 == 0), the wave-span form of pool4_kernel: a wave owns 64 consecutive lookups, finds
// the run heads among them with one ballot, its four 16-lane groups take the heads round-robin; a group fetches 16 map
// entries at once and keeps 16 row loads in flight.  (One group per lookup with only the heads working keeps one group in
// twenty busy at 20 lookups per bag: 2.13 M lookups 416 -> see profiles.)  Same sums in the same (index) order.
__global__ __launch_bounds__(kThreads) void pool_gather4_kernel(int N, int B, int D4, const int64_t* __restrict__ rowidx,
                                                               const int64_t* __restrict__ tableidx,
                                                               const int* __restrict__ uid, const float4* __restrict__ rows,
                                                               const float* __restrict__ psw, float4* __restrict__ out) {
  __shared__ unsigned char hpos[kThreads / kWave][kWave];
  const int w = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  const int n0 = (blockIdx.x * (kThreads / kWave) + w) * kWave;
  if (n0 >= N) return;  // (wave-uniform)
  const int n = n0 + lane;
  bool head = false;
  if (n < N) head = n == 0 || rowidx[n - 1] != rowidx[n] || tableidx[n - 1] != tableidx[n];
  const unsigned long long heads = __ballot(head);
  if (head) hpos[w][__popcll(heads & ((1ull << lane) - 1ull))] = (unsigned char)lane;
  const int nheads = __popcll(heads);
  const int g = lane >> 4, l = lane & 15, sh = lane & 48;
  for (int k = g; k < nheads; k += 4) {
    const int pos = hpos[w][k];
    const int hn = n0 + pos;
    const int64_t r = rowidx[hn], tb = tableidx[hn];
    int sl;
    if (k + 1 < nheads) {
      sl = hpos[w][k + 1] - pos;
    } else {  // the span's last run may go on behind it: 16 candidates per ballot (one group gets here)
      sl = min(N, n0 + kWave) - hn;
      if (hn + sl < N)
        for (;;) {
          const int c = hn + sl + l;
          const bool same = c < N && rowidx[c] == r && tableidx[c] == tb;
          const unsigned m = (unsigned)(__ballot(!same) >> sh) & 0xffffu;
          if (m) { sl += __builtin_ctz(m); break; }
          sl += 16;
        }
    }
    float4* o = out + ((size_t)tb * B + r) * D4;
    for (int e0 = 0; e0 < D4; e0 += 16) {
      const int e = e0 + l;
      const bool ev = e < D4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ev) acc = o[e];
      for (int base = 0; base < sl; base += 16) {
        const int j = base + l;
        const int u = j < sl ? uid[hn + j] : 0;
        const float wv = (psw && j < sl) ? psw[hn + j] : 1.f;
        const int cnt = min(16, sl - base);
        float4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int uq = __shfl(u, q, 16);
          if (q < cnt && ev) v[q] = rows[(size_t)uq * D4 + e];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {  // added in index order
          const float wq = __shfl(wv, q, 16);
          if (q < cnt && ev) {
            if (psw) vfma(acc, wq, v[q]);
            else vadd(acc, v[q]);
          }
        }
      }
      if (ev) o[e] = acc;
    }
  }
}

// Gu[u, :] = sum over the occurrences n of distinct pair u of (psw[n] *) d_output[table(n), row(n), :], in a fixed order.
// The map lists the occurrences sorted by pair (occ[], a pair's in index order).  That list is cut into SLICES of
// kGsSlice positions, whatever pairs they belong to: gsum_slice_kernel gives every slice to one 16-lane group, which
// fetches its gradient rows (16 in flight, the index chain occ -> bag row paid once per slice) and adds them in
// order; a pair whose run lies inside the slice is finished and stored, the part of a run that began in an earlier slice
// goes to P[slice][0], the part of a run that continues into the next one to P[slice][1].  gsum_fold_kernel then
// finishes the pairs that span slices: the group of the slice a run ENDS in adds P[first][1], .., P[last - 1][1], P[last][0]
// in slice order.  Every group does the same amount of work whatever the skew -- on a Zipf stream 70 % of the
// occurrences belong to pairs hit 16+ times, which the former one-owner-per-pair kernel summed with a work-group each,
// one after the other -- and the order of addition is fixed by the sort: deterministic.  Slice length measured at 2.1 M /
// 327k lookups (Zipf 1.2): 128 -> 208 / 73 us, 64 -> 152 / 46, 32 -> 127 / 28 (the fold grows the other way: 13 / 29 / 57 us
// before long runs were folded by the whole work-group, 17 / 7 us with it); the one-owner kernel took 427 / 115 us.
constexpr int kGsSlice = 32;
constexpr int kGsRounds = kGsSlice / 16;
constexpr int kGsThreads = 256;
__device__ __forceinline__ long long shfl64(long long v, int j) {
  return ((long long)__shfl((int)(v >> 32), j, 16) << 32) | (unsigned)__shfl((int)v, j, 16);
}
template <typename V>
__global__ __launch_bounds__(kGsThreads) void gsum_slice_kernel(DedupMap M, int N, int B, int DV, const int64_t* __restrict__ rowidx,
                                                               const int64_t* __restrict__ tableidx,
                                                               const float* __restrict__ psw, const V* __restrict__ dout,
                                                               V* __restrict__ Gu, V* __restrict__ P) {
  const int l = threadIdx.x & 15;
  const int s = blockIdx.x * (kGsThreads / 16) + threadIdx.x / 16;
  const int k0 = s * kGsSlice;
  if (k0 >= N) return;  // (no barrier below)
  const int kend = min(N, k0 + kGsSlice);
  int n[kGsRounds];
#pragma unroll
  for (int r = 0; r < kGsRounds; ++r) {
    const int k = k0 + 16 * r + l;
    n[r] = k < kend ? M.occ[k] : -1;
  }
  const int nprev = k0 > 0 ? M.occ[k0 - 1] : -1, nnext = kend < N ? M.occ[kend] : -1;
  long long off[kGsRounds];
  float w[kGsRounds];
  int up[kGsRounds];
#pragma unroll
  for (int r = 0; r < kGsRounds; ++r) {
    off[r] = 0, w[r] = 1.f, up[r] = -1;
    if (n[r] >= 0) {
      off[r] = (tableidx ? (long long)tableidx[n[r]] * B : 0ll) + rowidx[n[r]];  // (64-bit: tables * B may exceed 2^31)
      up[r] = M.uid[n[r]];
      if (psw) w[r] = psw[n[r]];
    }
  }
  const int uprev = nprev >= 0 ? M.uid[nprev] : -1, unext = nnext >= 0 ? M.uid[nnext] : -2;
  unsigned last = 0;  // bit r: position (r, l) is the last of its pair's run
#pragma unroll
  for (int r = 0; r < kGsRounds; ++r) {
    int nx = __shfl_down(up[r], 1, 16);
    const int first_of_next = r + 1 < kGsRounds ? __shfl(up[r + 1 < kGsRounds ? r + 1 : r], 0, 16) : unext;
    if (l == 15) nx = first_of_next;
    if (k0 + 16 * r + l + 1 >= kend) nx = unext;
    if (nx != up[r]) last |= 1u << r;
  }
  const bool began_before = __shfl(up[0], 0, 16) == uprev;
  V* Ps = P + (size_t)s * 2 * DV;
  for (int eb = 0; eb < DV; eb += 16) {
    const int e = eb + l;
    const bool ev = e < DV;
    V acc;
    vzero(acc);
    bool head = began_before, open = false;
#pragma unroll
    for (int r = 0; r < kGsRounds; ++r) {
      const int cnt = min(16, kend - (k0 + 16 * r));
      if (cnt <= 0) break;
      V v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const long long oj = shfl64(off[r], j);
        if (j < cnt && ev) v[j] = dout[(size_t)oj * DV + e];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float wj = __shfl(w[r], j, 16);
        const int uj = __shfl(up[r], j, 16);
        const bool lj = (__shfl((int)last, j, 16) >> r) & 1;
        if (j < cnt) {
          if (ev) {
            if (psw) vfma(acc, wj, v[j]);
            else vadd(acc, v[j]);
          }
          open = true;
          if (lj) {
            if (ev) {
              if (head) Ps[e] = acc;
              else Gu[(size_t)uj * DV + e] = acc;
            }
            vzero(acc);
            head = false, open = false;
          }
        }
      }
    }
    if (open && ev) Ps[DV + e] = acc;
  }
}

// partials [a, b) of the run that began in slice sf and ends in slice s (partial i: P[sf + i][1], the last one P[s][0]),
// column e, added in order, 16 loads in flight
template <typename V>
__device__ __forceinline__ V fold_partials(const V* __restrict__ P, int sf, int s, int a, int b, int DV, int e, bool ev) {
  V acc;
  vzero(acc);
  for (int base = a; base < b; base += 16) {
    const int c = min(16, b - base);
    V v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int t = sf + base + j;
      if (j < c && ev) v[j] = P[((size_t)t * 2 + (t == s ? 0 : 1)) * DV + e];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < c && ev) vadd(acc, v[j]);
  }
  return acc;
}

// One 16-lane group per slice: the group of the slice a run ENDS in folds that run's partials.  A run of more than
// kGsFoldCoop partials (a hot row of a skewed stream: thousands of occurrences) is folded by the 16 groups of the
// work-group together -- group g sums the g-th part, the part sums are added in group order through LDS -- instead of
// one group walking hundreds of rows 16 at a time while the launch waits for it.  Fixed partition, fixed order.
constexpr int kGsFoldCoop = 64;
template <typename V>
__global__ __launch_bounds__(kGsThreads) void gsum_fold_kernel(DedupMap M, int N, int DV, const V* __restrict__ P, V* __restrict__ Gu) {
  constexpr int kG = kGsThreads / 16;
  __shared__ V part[kG][16];
  __shared__ int longs[kG][3];
  __shared__ int nlong;
  const int l = threadIdx.x & 15, g = threadIdx.x / 16;
  const int s = blockIdx.x * kG + g;
  const int k0 = s * kGsSlice;
  if (threadIdx.x == 0) nlong = 0;
  __syncthreads();
  int u = -1, sf = 0;
  if (k0 < N && k0 > 0) {
    u = M.uid[M.occ[k0]];
    if (M.uid[M.occ[k0 - 1]] != u) u = -1;  // no run crosses into this slice
    else {
      const int lo = M.occ_off[u], hi = M.occ_off[u + 1];
      if ((hi - 1) / kGsSlice != s) u = -1;  // it ends in a later slice, whose group folds it
      else sf = lo / kGsSlice;
    }
  }
  if (u >= 0 && s - sf + 1 > kGsFoldCoop) {
    if (l == 0) {
      const int i = atomicAdd(&nlong, 1);
      longs[i][0] = u, longs[i][1] = sf, longs[i][2] = s;
    }
    u = -1;
  }
  if (u >= 0)
    for (int eb = 0; eb < DV; eb += 16) {
      const int e = eb + l;
      const V acc = fold_partials<V>(P, sf, s, 0, s - sf + 1, DV, e, e < DV);
      if (e < DV) Gu[(size_t)u * DV + e] = acc;
    }
  __syncthreads();
  const int nl = nlong;
  for (int h = 0; h < nl; ++h) {  // (work-group-uniform; the order of the list does not touch any value)
    const int uu = longs[h][0], f = longs[h][1], se = longs[h][2];
    const int cnt = se - f + 1, per = (cnt + kG - 1) / kG;
    const int a = min(cnt, g * per), b = min(cnt, a + per);
    for (int eb = 0; eb < DV; eb += 16) {
      const int e = eb + l;
      part[g][l] = fold_partials<V>(P, f, se, a, b, DV, e, e < DV);
      __syncthreads();
      if (g == 0 && e < DV) {
        V acc = part[0][l];
        for (int q = 1; q < kG; ++q) vadd(acc, part[q][l]);
        Gu[(size_t)uu * DV + e] = acc;
      }
      __syncthreads();
    }
  }
}

// gradient with respect to nn.EmbeddingBag's per_sample_weights: out[bag] = sum_n w_n row_n  ->  dL/dw_n = <d_out[bag(n)], row_n>.
// One 16-lane group per lookup (rows are the forward's, kept by ttx_tt_forward_wr).
__global__ __launch_bounds__(kThreads) void psw_grad_kernel(int N, int B, int D, const float* __restrict__ rows,
                                                           const int64_t* __restrict__ rowidx,
                                                           const int64_t* __restrict__ tableidx,
                                                           const float* __restrict__ dout, float* __restrict__ d_psw) {
  const int n = blockIdx.x * (kThreads / 16) + threadIdx.x / 16;
  const int l = threadIdx.x & 15;
  float acc = 0.f;
  if (n < N) {
    const float* r = rows + (size_t)n * D;
    const float* g = dout + ((size_t)tableidx[n] * B + rowidx[n]) * D;
    for (int e = l; e < D; e += 16) acc = fmaf(r[e], g[e], acc);
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 16);
  if (n < N && l == 0) d_psw[n] = acc;
}

__device__ __forceinline__ float apply_one(int optim, float g, float w, float lr, float eps, float* st) {
  if (optim == TTX_OPTIM_SGD) return w - lr * g;
  const float s = *st + g * g;
  *st = s;
  return w - lr * g / (sqrtf(s) + eps);
}

// one work-group per core slice: sum the slice's partials in a fixed order and
// apply.  DENSE writes the gradient (zeros for untouched slices: no memset of
// d_tt_cores is needed); SGD / ADAGRAD skip untouched slices (g == 0).
// Short slices with many partial rows: the 256 threads form G = 256/V groups
// (V = lanes covering the slice, 4 floats per lane); group g sums partials g,
// g+G, .. in index order, the G group sums are then added in group order
// through LDS.  Long slices: 4 floats per thread, partials summed in order.
constexpr int kReduceThreads = 1024;  // upper bound; launched with 512 when the largest slice is <= 4096 floats
static size_t reduce_lds_bytes(int threads) { return (size_t)threads * (sizeof(float4) + sizeof(int)); }  // red[] + s_hot[]
constexpr int kSegPivot = 32;         // chunk partials per segment of the pivot core
// A slice is HOT when it holds more than two segments' worth of partials (a skewed index stream puts
// a third of a batch on one slice): one work-group per SEGMENT then sums its share, and the last one
// to arrive folds the segment sums in segment order and applies the optimizer -- all other slices keep
// their single owner.  Deterministic either way (fixed partition, fixed order).
__device__ __forceinline__ int seg_len(int t) { return t == 1 ? kSegPivot : kSegThin; }
// The PIVOT core's hot slices (few rows, 4096 floats each) are split by COLUMNS instead: work-group j takes float4
// columns [j C, (j + 1) C) of every hot slice, its threads form nthreads / C row groups, group g sums rows g, g + G,
// .. (eight in flight), the group sums are folded in group order through LDS and the optimizer is applied to those
// columns -- no second pass, no arrival counter, no fence.  (Row segments of 32 chunk partials plus the last
// arriver's fold of the segment sums were the tail of the launch on a skewed stream: 16 + 6 us of serial load
// rounds.  The thin cores keep row segments: many rows of 128 floats -- column blocks of 64 bytes would leave
// eight work-groups to read 4.7 MB, measured 43 us.)
#ifndef TTX_HOT_COLS
#define TTX_HOT_COLS 8
#endif
#ifndef TTX_HOT_NF
#define TTX_HOT_NF 4
#endif
#ifndef TTX_RTHREADS_SMALL
#define TTX_RTHREADS_SMALL 512
#endif
constexpr int kHotNF = TTX_HOT_NF;  // rows in flight per lane on the hot-slice paths
constexpr int kHotColsPivot = TTX_HOT_COLS;  // (kHotRowsPivot, kSegThin: ttx_internal.h)
#ifndef TTX_MAX_HOT_PIVOT
#define TTX_MAX_HOT_PIVOT 8
#endif
constexpr int kMaxHotPivot = TTX_MAX_HOT_PIVOT;  // hot pivot slices beyond which their owners keep them (reduce_apply_kernel)
// column work-groups of the launch: one per block of kHotColsPivot float4 columns, each walks ALL hot slices of the pivot
// (a launch without hot slices pays for V / C work-groups that look at the offsets once and leave)
__host__ __device__ __forceinline__ int hot_wgs(int total_max, int sl, int t) {
  return (sl & 3) ? 0 : (sl / 4 + kHotColsPivot - 1) / kHotColsPivot;
}

struct ApplyEmit {
  int optim;
  float lr, eps;
  float *wt, *stt, *dw;  // already offset to the slice
  struct Pre { float4 w, s; };
  // the weights (and Adagrad state) of lane v are fetched BEFORE the partial rows are summed: one memory
  // round trip less on the critical path of a latency-bound kernel
  __device__ __forceinline__ Pre prefetch(int v) const {
    Pre p{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if (optim != TTX_OPTIM_DENSE) {
      p.w = *(const float4*)(wt + (size_t)v * 4);
      if (optim == TTX_OPTIM_ADAGRAD) p.s = *(const float4*)(stt + (size_t)v * 4);
    }
    return p;
  }
  __device__ __forceinline__ void operator()(int v, float4 acc, Pre p) const {
    const size_t o = (size_t)v * 4;
    if (optim == TTX_OPTIM_DENSE) {
      *(float4*)(dw + o) = acc;
    } else {
      float4 wv = p.w, sv = p.s;
      wv.x = apply_one(optim, acc.x, wv.x, lr, eps, &sv.x);
      wv.y = apply_one(optim, acc.y, wv.y, lr, eps, &sv.y);
      wv.z = apply_one(optim, acc.z, wv.z, lr, eps, &sv.z);
      wv.w = apply_one(optim, acc.w, wv.w, lr, eps, &sv.w);
      *(float4*)(wt + o) = wv;
      if (optim == TTX_OPTIM_ADAGRAD) *(float4*)(stt + o) = sv;
    }
  }
};
struct StoreEmit {
  float* dst;
  struct Pre {};
  __device__ __forceinline__ Pre prefetch(int) const { return Pre{}; }
  __device__ __forceinline__ void operator()(int v, float4 acc, Pre) const { ((float4*)dst)[v] = acc; }
};

// sum rows row(beg) .. row(end-1) of `pc` (sl floats each, sl % 4 == 0) in a fixed order and hand every
// float4 lane of the result to emit(v, sum).  Whole work-group; contains block barriers.
template <class RowFn, class Emit, bool DEEP = false>
__device__ __forceinline__ void sum_rows4(const float* __restrict__ pc, int sl, int beg, int end, float4* red,
                                          RowFn row, Emit emit) {
  const int nthreads = blockDim.x, tid = threadIdx.x;
  const int V = sl / 4, cnt = end - beg;
  if (V <= nthreads / 2 && cnt >= 4) {
    // G groups of V lanes share the rows; at least 8 rows per group, so a short list stays with few
    // groups (cheap final fold) and a long one gets every thread
    int G = nthreads / V;
    if (G > cnt / 8) G = cnt / 8 > 0 ? cnt / 8 : 1;
    const int g = tid / V, v = tid - g * V;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    typename Emit::Pre pre{};
    if (g == 0) pre = emit.prefetch(v);
    if (g < G) {
      int i = beg + g;
      if (DEEP || cnt >= 16 * G) {  // many rows per group (a hot slice's segment, a warm slice's owner): the load rounds ARE the time
        for (; i + (kHotNF - 1) * G < end; i += kHotNF * G) {
          float4 x[kHotNF];
#pragma unroll
          for (int u = 0; u < kHotNF; ++u) x[u] = ((const float4*)(pc + row(i + u * G) * sl))[v];
#pragma unroll
          for (int u = 0; u < kHotNF; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
        }
      }
      for (; i + 3 * G < end; i += 4 * G) {  // four rows in flight per lane (eight: slower, 13.7 vs 12.3 us at cfg2)
        const size_t r0 = row(i), r1 = row(i + G), r2 = row(i + 2 * G), r3 = row(i + 3 * G);
        const float4 x0 = ((const float4*)(pc + r0 * sl))[v], x1 = ((const float4*)(pc + r1 * sl))[v];
        const float4 x2 = ((const float4*)(pc + r2 * sl))[v], x3 = ((const float4*)(pc + r3 * sl))[v];
        acc.x += x0.x; acc.y += x0.y; acc.z += x0.z; acc.w += x0.w;
        acc.x += x1.x; acc.y += x1.y; acc.z += x1.z; acc.w += x1.w;
        acc.x += x2.x; acc.y += x2.y; acc.z += x2.z; acc.w += x2.w;
        acc.x += x3.x; acc.y += x3.y; acc.z += x3.z; acc.w += x3.w;
      }
      for (; i < end; i += G) {
        const float4 x = ((const float4*)(pc + row(i) * sl))[v];
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
      red[tid] = acc;
    }
    __syncthreads();
    if (g == 0) {
      for (int k = 1; k < G; ++k) {
        const float4 x = red[k * V + v];
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
      emit(v, acc, pre);
    }
    return;
  }
  for (int v = tid; v < V; v += nthreads) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const typename Emit::Pre pre = emit.prefetch(v);
    int i = beg;
    for (; i + 3 < end; i += 4) {  // four rows in flight
      const size_t r0 = row(i), r1 = row(i + 1), r2 = row(i + 2), r3 = row(i + 3);
      const float4 x0 = ((const float4*)(pc + r0 * sl))[v], x1 = ((const float4*)(pc + r1 * sl))[v];
      const float4 x2 = ((const float4*)(pc + r2 * sl))[v], x3 = ((const float4*)(pc + r3 * sl))[v];
      acc.x += x0.x; acc.y += x0.y; acc.z += x0.z; acc.w += x0.w;
      acc.x += x1.x; acc.y += x1.y; acc.z += x1.z; acc.w += x1.w;
      acc.x += x2.x; acc.y += x2.y; acc.z += x2.z; acc.w += x2.w;
      acc.x += x3.x; acc.y += x3.y; acc.z += x3.z; acc.w += x3.w;
    }
    for (; i < end; ++i) {
      const float4 x = ((const float4*)(pc + row(i) * sl))[v];
      acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
    emit(v, acc, pre);
  }
}

struct ListRow {  // thin cores: partial row = lookup id from the sorted order
  const int* list;
  __device__ __forceinline__ size_t operator()(int i) const { return (size_t)list[i]; }
};
struct IotaRow {  // pivot core: partial row = chunk slot
  __device__ __forceinline__ size_t operator()(int i) const { return (size_t)i; }
};
struct SegRow {  // final fold of a hot slice: segment j's slot (1 = the slice starts inside the segment)
  int j_first, first_slot;
  __device__ __forceinline__ size_t operator()(int i) const { return (size_t)(2 * (j_first + i) + (i == 0 ? first_slot : 0)); }
};

#include "ttx_cache_scatter.inc"
// the cache rows' SGD scatter riding in reduce_apply's launch (ttx_tt_backward_wc): work-groups [first, first + nmain + hot) of the
// grid run cache_scatter_add_body on their first kScatterThreads threads; dst == NULL: none
struct CacheTail {
  int first, N, D, nmain, K;
  float scale;
  const int* skip_dev;
  const float* grad;
  const int32_t* loc;
  const int64_t* rowidx;
  float* dst;
};
#ifndef TTX_REDUCE_WAVES
#define TTX_REDUCE_WAVES 6
#endif
// (second launch bound = waves per SIMD the register allocation has to leave room for: the hot-slice paths pushed the
//  kernel from 77 to 89-99 VGPRs, i.e. from three to two 512-thread work-groups per CU, +1.5 us at the benchmark batch)
// element t of a small array that lives in the kernel arguments, as a chain of selects: a RUN-TIME subscript of a by-value
// struct member sends the whole struct through scratch memory (44 bytes per lane here) -- a trip to memory in front of the first
// real one (round 6: reduce_apply's slice owners read five such members before their offsets)
template <class T>
__device__ __forceinline__ T sel_core(const T (&a)[TTX_MAX_CORES], int t) {
  T v = a[0];
#pragma unroll
  for (int c = 1; c < TTX_MAX_CORES; ++c) v = (t == c) ? a[c] : v;
  return v;
}
// (slice number within the launch -> core t, slice within the core)
__device__ __forceinline__ int core_of_slice(const Dims& d, int& b) {
  int t = 0;
#pragma unroll
  for (int c = 0; c < TTX_MAX_CORES - 1; ++c)
    if (t == c && c < d.T - 1 && b >= d.S[c]) { b -= d.S[c]; t = c + 1; }
  return t;
}
__global__ __launch_bounds__(kReduceThreads, TTX_REDUCE_WAVES) void reduce_apply_kernel(Dims d, Plan P, Partials PC,
                                                               int optim, float lr, float eps,
                                                               CorePtrs W, CorePtrs St,
                                                               CorePtrs DW, int nslices, int rows_max, CacheTail CT, int pack) {
  if (CT.dst && (int)blockIdx.x >= CT.first) {  // (work-group-uniform) a work-group of the cache rows' scatter
    // (all threads: the body's hot-row path has work-group barriers; threads beyond kScatterThreads only wait at them)
    cache_scatter_add_body((int)blockIdx.x - CT.first, CT.N, CT.D, CT.scale, CT.skip_dev, CT.grad, CT.loc, CT.rowidx, CT.dst, CT.nmain, CT.K);
    return;
  }
  // (round 5) sized by the launch (blockDim.x float4 + blockDim.x int: reduce_lds_bytes): 20 KB of static LDS for 1024 threads held a
  // launch of 128-thread work-groups to eight per CU
  extern __shared__ __attribute__((aligned(16))) float4 ra_lds[];
  float4* red = ra_lds;
  int* s_hot = (int*)(ra_lds + blockDim.x);
  __shared__ int s_last, s_nhot;
  const int nthreads = blockDim.x, tid = threadIdx.x;
  int b = blockIdx.x;
  const bool skip_pivot = rows_max < 0;  // (ablation only)
  if (rows_max < 0) rows_max = -rows_max;
  if (pack > 1) {
    // ---------------- (round 6) small slices, many of them: ONE WAVE per slice, `pack` slices per work-group ----------------
    // Two cores over 11 M rows are 2 x 3317 slices of 256 floats with ~3 partial rows each: a work-group per slice was 6634
    // work-groups whose 64 float4 lanes of work sat behind a work-group barrier and an LDS fold.  A slice of at most 64 float4
    // lanes is one wave's: lane v sums the slice's rows in index order (every request of a round of eight in flight), no
    // barrier, no LDS, a quarter of the work-groups.  Hot slices stay with the segment work-groups, as below.
    const int npk = (nslices + pack - 1) / pack;
    if (b < npk) {
      const int lane = tid & (kWave - 1);
      int bs = b * pack + tid / kWave;
      if (bs >= nslices) return;
      const int t = core_of_slice(d, bs);
      const int s = bs, sl = sel_core(d.slice, t), V = sl / 4;
      if (skip_pivot && t == 1) return;
      const int* offs = (t == 1) ? P.chunk_off : sel_core(P.off, t);
      const int beg = offs[s], end = offs[s + 1];
      float* const wt_ = sel_core(W.c, t);
      float* const st_ = sel_core(St.c, t);
      float* const dw_ = sel_core(DW.c, t);
      const size_t base = (size_t)s * sl;
      if (beg == end) {
        if (optim == TTX_OPTIM_DENSE)
          for (int e = lane; e < sl; e += kWave) dw_[base + e] = 0.f;
        return;
      }
      if (end - beg > (t == 1 ? kHotRowsPivot : 2 * seg_len(t)) && PC.hot_cnt && !(t == 1 && P.hdr[8 + 1] > kMaxHotPivot)) return;
      if (lane >= V) return;
      const ApplyEmit ap{optim, lr, eps, wt_ + base, st_ ? st_ + base : nullptr, dw_ ? dw_ + base : nullptr};
      const ApplyEmit::Pre pre = ap.prefetch(lane);
      const float4* rows = (const float4*)(sel_core(PC.pc, t)) + lane;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int i = beg;
      for (; i + 7 < end; i += 8) {
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = rows[(size_t)(i + u) * V];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
      }
      if (i < end) {  // the last round: rows behind the end read the last row again and are not added
        float4 x[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) x[u] = rows[(size_t)min(i + u, end - 1) * V];
#pragma unroll
        for (int u = 0; u < 7; ++u)
          if (i + u < end) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
      }
      ap(lane, acc, pre);
      return;
    }
    b = b - npk + nslices;  // a segment work-group: numbered behind the slices, as without packing
  }
  if (b >= nslices) {
    // ---------------- segment work-group (t, j): its share of the hot slice(s) it intersects ----------------
    b -= nslices;
    int t = 0;
    for (;; ++t) {
      const int nseg = (t == 1) ? hot_wgs(P.max_chunks, d.slice[1], 1) : (rows_max + seg_len(t) - 1) / seg_len(t);
      if (b < nseg || t == d.T - 1) break;
      b -= nseg;
    }
    if (P.hdr[8 + t] == 0) return;  // the plan kernel found no hot slice in this core (-1: it did not look)
    // (round 5) MANY hot pivot slices -- a large batch over few slices: p_1 = 58 at 327k lookups makes all 232 of them hot -- stay
    // with their owners: every column work-group walks EVERY hot slice, one barrier-bounded round per slice, so the launch took
    // 919 us there (t4big) for 170 MB of partials.  The column split is for the few hot slices of a skewed stream.
    if (t == 1 && P.hdr[8 + 1] > kMaxHotPivot) return;
    if (t == 1) {  // ---- pivot: column work-group j -- float4 columns [j C, (j + 1) C) of EVERY hot slice ----
      const int sl = d.slice[1], V = sl / 4, C = kHotColsPivot;
      const int c0 = b * C, c1 = min(V, c0 + C);
      if (c0 >= c1) return;
      const int* off = P.chunk_off;
      const int S = d.S[1];
      const int G = nthreads / C, g = tid / C, v = c0 + (tid - g * C);
      const bool act = g < G && v < c1;
      const float* __restrict__ pc = PC.pc[1];
      for (int s0 = 0; s0 < S; s0 += nthreads) {  // (work-group-uniform control flow throughout)
        if (tid == 0) s_nhot = 0;
        __syncthreads();
        const int sidx = s0 + tid;
        if (sidx < S && off[sidx + 1] - off[sidx] > kHotRowsPivot) s_hot[atomicAdd(&s_nhot, 1)] = sidx;
        __syncthreads();
        const int nh = s_nhot;
        for (int h = 0; h < nh; ++h) {  // (the order in which a work-group takes its hot slices does not matter)
          const int sh = s_hot[h];
          const int beg = off[sh], end = off[sh + 1];
          const size_t base = (size_t)sh * sl;
          const ApplyEmit ap{optim, lr, eps, W.c[1] + base, St.c[1] ? St.c[1] + base : nullptr, DW.c[1] ? DW.c[1] + base : nullptr};
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          typename ApplyEmit::Pre pre{};
          if (act && g == 0) pre = ap.prefetch(v);
          if (act) {
            for (int i = beg + g; i < end; i += kHotNF * G) {  // kHotNF rows in flight per lane (the tail predicated, not serialised)
              float4 x[kHotNF];
#pragma unroll
              for (int u = 0; u < kHotNF; ++u) {
                const int r = i + u * G;
                x[u] = r < end ? ((const float4*)(pc + (size_t)r * sl))[v] : make_float4(0.f, 0.f, 0.f, 0.f);
              }
#pragma unroll
              for (int u = 0; u < kHotNF; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
            }
            red[tid] = acc;
          }
          __syncthreads();
          if (act && g == 0) {
            for (int q = 1; q < G; ++q) {
              const float4 x = red[q * C + (v - c0)];
              acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
            }
            ap(v, acc, pre);
          }
          __syncthreads();
        }
      }
      return;
    }
    const int SEG = seg_len(t), sl = sel_core(d.slice, t), St_ = sel_core(d.S, t);  // (selects, not subscripts: sel_core)
    if ((sl & 3) != 0) return;  // (odd slice sizes stay with their single owner)
    const int total = (t == 1) ? P.hdr[0] : P.hdr[2];
    const int p0 = b * SEG, p1 = min(p0 + SEG, total);
    if (p0 >= total) return;
    const int* off = (t == 1) ? P.chunk_off : sel_core(P.off, t);
    // the offset table goes to LDS in one coalesced round when it fits (a segment work-group of a
    // uniform stream has nothing to do and should find that out in one memory round trip, not ten)
    int* soff = (int*)red;
    const bool in_lds = St_ + 1 <= (int)(nthreads * sizeof(float4) / sizeof(int));
    if (in_lds) {
      for (int e = tid; e <= St_; e += nthreads) soff[e] = off[e];
      __syncthreads();
      off = soff;
    }
    int lo = 0, hi = St_;  // slice containing position p0: the last s with off[s] <= p0
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (off[mid] <= p0) lo = mid; else hi = mid;
    }
    bool any_hot = false;  // (wave-uniform: every thread scans the same few slices)
    for (int s = lo; s < St_ && off[s] < p1; ++s) any_hot |= (off[s + 1] - off[s] > 2 * SEG);
    if (!any_hot) return;
    if (in_lds) {  // `red` is about to be used by the reductions: back to the global table
      __syncthreads();
      off = (t == 1) ? P.chunk_off : sel_core(P.off, t);
    }
    const float* __restrict__ pc = sel_core(PC.pc, t);
    for (int s = lo; s < St_; ++s) {  // at most two hot slices can touch one segment
      const int sbeg = off[s], send = off[s + 1];
      if (sbeg >= p1) break;
      if (send - sbeg <= 2 * SEG) continue;  // not hot: its owner does everything
      const int beg = max(sbeg, p0), end = min(send, p1);
      const int slot = (sbeg > p0) ? 1 : 0;
      float* dst = sel_core(PC.seg, t) + (size_t)(2 * b + slot) * sl;
      if (t == 1) sum_rows4(pc, sl, beg, end, red, IotaRow{}, StoreEmit{dst});
      else sum_rows4<IotaRow, StoreEmit, true>(pc, sl, beg, end, red, IotaRow{}, StoreEmit{dst});  // (partials lie in sorted order)
      // arrival (the "last block" pattern).  Producer: the work-group's plain stores are complete at the barrier
      // (work-group scope), then ONE lane writes the XCD's dirty L2 lines back (agent-scope release: buffer_wbl2
      // works on the cache, not on a thread's own stores) and counts the work-group in; the work-group that
      // completes the count folds the segment sums after ONE lane's agent-scope acquire (invalidates this CU's L1).
      // (Every thread calling __threadfence() -- 512 release + acquire pairs per work-group -- cost 2-4x one lane's,
      // MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility".)
      __syncthreads();
      if (tid == 0) {
        int idx = s;
        #pragma unroll
        for (int tt = 0; tt < TTX_MAX_CORES - 1; ++tt) idx += tt < t ? d.S[tt] : 0;
        const int j_first = sbeg / SEG, j_last = (send - 1) / SEG;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int old = __hip_atomic_fetch_add(&PC.hot_cnt[idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old == j_last - j_first);
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      if (s_last) {
        const int j_first = sbeg / SEG, j_last = (send - 1) / SEG;
        const size_t base = (size_t)s * sl;
        const ApplyEmit ap{optim, lr, eps, sel_core(W.c, t) + base, sel_core(St.c, t) ? sel_core(St.c, t) + base : nullptr,
                           sel_core(DW.c, t) ? sel_core(DW.c, t) + base : nullptr};
        sum_rows4(sel_core(PC.seg, t), sl, 0, j_last - j_first + 1, red, SegRow{j_first, sbeg > j_first * SEG ? 1 : 0}, ap);
      }
      __syncthreads();
    }
    return;
  }
  // ---------------- slice owner ----------------
  const int t = core_of_slice(d, b);
  const int s = b;
  const int sl = sel_core(d.slice, t);
  if (skip_pivot && t == 1) return;
  int beg, end;
  const int* __restrict__ list;
  if (t == 1) { beg = P.chunk_off[s]; end = P.chunk_off[s + 1]; list = nullptr; }
  else { const int* offs = sel_core(P.off, t); beg = offs[s]; end = offs[s + 1]; list = nullptr; }  // partial rows = positions of the sorted order
  const size_t base = (size_t)s * sl;
  float* __restrict__ dw = sel_core(DW.c, t);
  float* __restrict__ wt = sel_core(W.c, t);
  float* __restrict__ stt = sel_core(St.c, t);
  if (beg == end) {
    if (optim == TTX_OPTIM_DENSE)
      for (int e = tid; e < sl; e += nthreads) dw[base + e] = 0.f;
    return;
  }
  const float* __restrict__ pc = sel_core(PC.pc, t);
  if ((sl & 3) == 0) {
    if (end - beg > (t == 1 ? kHotRowsPivot : 2 * seg_len(t)) && PC.hot_cnt && !(t == 1 && P.hdr[8 + 1] > kMaxHotPivot))
      return;  // hot: the segment / column work-groups own it (unless the pivot has too many hot slices for the column split)
    const ApplyEmit ap{optim, lr, eps, wt + base, stt ? stt + base : nullptr, dw ? dw + base : nullptr};
    if (list) sum_rows4(pc, sl, beg, end, red, ListRow{list}, ap);
    else sum_rows4(pc, sl, beg, end, red, IotaRow{}, ap);
    return;
  }
  for (int e = tid; e < sl; e += nthreads) {
    float g = 0.f;
    for (int i = beg; i < end; ++i) g += pc[(list ? (size_t)list[i] : (size_t)i) * sl + e];
    if (optim == TTX_OPTIM_DENSE) {
      dw[base + e] = g;
    } else {
      float sv = optim == TTX_OPTIM_ADAGRAD ? stt[base + e] : 0.f;
      wt[base + e] = apply_one(optim, g, wt[base + e], lr, eps, &sv);
      if (optim == TTX_OPTIM_ADAGRAD) stt[base + e] = sv;
    }
  }
}

#include "ttx_tt_spec.inc"

static bool spec_shape(const Dims& d) { return spec_match(d) != SPEC_NONE; }
// lookups per chunk of the specialised kernels (Shape3::MC of the variant that runs)
// Large batches of the r <= 32 shapes: four sub-chunks per plan chunk -- one pivot partial per 64 lookups
// (measured at 327k lookups: 1 / 2 / 4 / 8 sub-chunks -> 0.794 / 0.771 / 0.758 / 0.770 ms per step; at 82k
// lookups one sub-chunk wins, 0.227 vs 0.240, at 164k four, 0.460 vs 0.464: the switch sits at 128k).
#ifndef TTX_SUBCHUNK_NNZ
#define TTX_SUBCHUNK_NNZ 131072
#endif
#ifndef TTX_SUBCHUNKS
#define TTX_SUBCHUNKS 4
#endif
static int spec_mc(const Dims& d, long long nnz) {
  const int ks = nnz >= TTX_SUBCHUNK_NNZ ? TTX_SUBCHUNKS : 1;
  // (test build, ttx_debug_bwd32: the 32-lookup sub-chunks of bwd32_kernel want longer chunks)
  const int b32_mc = g_bwd32_mc;
  switch (spec_match(d)) {
    case SPEC_32_4_32_4: return (b32_mc && ks > 1) ? b32_mc : S_32_4_32_4::MC * (S_32_4_32_4::SUB ? ks : 1);
    case SPEC_16_4_16_4: return S_16_4_16_4::MC * (S_16_4_16_4::SUB ? ks : 1);
    case SPEC_32_4_32_8: return S_32_4_32_8::MC * (S_32_4_32_8::SUB ? ks : 1);
    case SPEC_16_4_16_8: return S_16_4_16_8::MC * (S_16_4_16_8::SUB ? ks : 1);
    case SPEC_64_4_64_8: return S_64_4_64_8::MC;
    case SPEC_64_4_64_4: return S_64_4_64_4::MC;
    case SPEC2_32_4_32_4: return S2_32_4_32_4::MC * (S2_32_4_32_4::SUB ? ks : 1);
    case SPEC2_16_4_16_4: return S2_16_4_16_4::MC * (S2_16_4_16_4::SUB ? ks : 1);
    case SPEC2_64_4_64_4: return S2_64_4_64_4::MC;
    case SPEC_32_8_32_8: return S_32_8_32_8::MC * (S_32_8_32_8::SUB ? ks : 1);
    case SPEC_64_8_64_8: return S_64_8_64_8::MC;
    case SPEC2_32_2_32_4: return S2_32_2_32_4::MC * (S2_32_2_32_4::SUB ? ks : 1);
    case SPEC2_64_2_64_4: return S2_64_2_64_4::MC;
    case SPEC2_16_2_16_4: return S2_16_2_16_4::MC * (S2_16_2_16_4::SUB ? ks : 1);
    case SPEC_16_8_16_8: return S_16_8_16_8::MC * (S_16_8_16_8::SUB ? ks : 1);
    case SPEC_128_4_128_4: return S_128_4_128_4::MC;
    case SPEC_128_4_128_8: return S_128_4_128_8::MC;
    case SPEC_128_8_128_8: return S_128_8_128_8::MC;
    case SPEC_32_8_32_16: return S_32_8_32_16::MC * (S_32_8_32_16::SUB ? ks : 1);
    case SPEC_32_16_32_16: return S_32_16_32_16::MC * (S_32_16_32_16::SUB ? ks : 1);
    case SPEC_16_16_16_16: return S_16_16_16_16::MC * (S_16_16_16_16::SUB ? ks : 1);
    case SPEC_64_16_64_16: return S_64_16_64_16::MC;
    case SPEC_32_16_32_32: return S_32_16_32_32::MC * (S_32_16_32_32::SUB ? ks : 1);
    case SPEC_16_16_16_32: return S_16_16_16_32::MC * (S_16_16_16_32::SUB ? ks : 1);
    case SPEC_32_4_32_32: return S_32_4_32_32::MC * (S_32_4_32_32::SUB ? ks : 1);
    case SPEC_16_4_16_32: return S_16_4_16_32::MC * (S_16_4_16_32::SUB ? ks : 1);
    case SPEC_32_8_32_32: return S_32_8_32_32::MC * (S_32_8_32_32::SUB ? ks : 1);
    case SPEC_16_8_16_32: return S_16_8_16_32::MC * (S_16_8_16_32::SUB ? ks : 1);
    case SPEC_16_8_16_16: return S_16_8_16_16::MC * (S_16_8_16_16::SUB ? ks : 1);
    case SPEC_64_8_64_16: return S_64_8_64_16::MC;
    default: return 0;
  }
}

// ---- two cores (round 4) ---------------------------------------------------------------------------------------------------
// A two-core lookup is ONE small product, row = core_0[i_0] [q0 x r1] * core_1[i_1] [r1 x q1]: 2 q0 r1 q1 multiply-adds on two
// slices of q0 r1 and r1 q1 floats -- 2 FLOP per byte fetched, nothing for a matrix pipe to win (cdna_hip_programming.md: byte
// work stays byte work).  What the generic kernels lose there (29 / 55 us forward / backward at the benchmark's batch, D = 64 =
// [8, 8], r = 32) is instructions: MFMA tiles padded from 8 to 16 columns, the block-walk bookkeeping.  These kernels do the
// least: a work-group per plan chunk (lookups of one core-1 slice) stages that slice once, TRANSPOSED, a wave stages one lookup's
// core-0 slice (and gradient row) with one coalesced load each and every lane produces its outputs from float4 LDS reads.
// Backward: d core_0 of the lookup goes out as its partial row, d core_1 accumulates in registers over the wave's lookups and
// the four waves' sums leave as the chunk's partial -- the layouts reduce_apply expects from any backward kernel.
// Shapes: r1 <= 128, q0, q1 <= 32 with r1 q1 <= 2048 (everything else stays on the generic kernels); r1 % 4 != 0 runs over zero-padded k tiles.
constexpr int kT2Threads = 256;
static bool t2_shape(const Dims& d) {
  // (round 5: q up to 32 -- the default two-core factorings of D = 320 .. 1024 are [16,20] .. [32,32] -- while core 1's slice is at most 2048
  //  floats: the backward keeps it in 32 registers per lane)
  return d.T == 2 && !g_disable_spec && d.r[1] <= 128 && d.q[0] <= 32 && d.q[1] <= 32 && d.r[1] * d.q[1] <= 2048;
}
struct T2Lds { int ldk, oBt, oA, oG, oR, floats; };  // ldk: row stride of the k-major tiles (r1 + 4: float4 rows, spread over the banks)
static T2Lds t2_lds(const Dims& d, bool bwd) {
  T2Lds L;
  L.ldk = (d.r[1] + 3) / 4 * 4 + 4;  // (r1 rounded up to whole float4s: the k loops run over zero padding when r1 % 4 != 0)
  L.oBt = 0;
  L.oA = L.oBt + d.q[1] * L.ldk;
  L.oG = L.oA + kWaves * d.q[0] * L.ldk;
  L.oR = L.oG + (bwd ? kWaves * (d.D + 4) : 0);
  L.floats = L.oR + (bwd ? kWaves * d.r[1] * d.q[1] : 0);
  return L;
}

// r1 % 4 != 0 (the reference tests' r = 13): columns r1 .. r1p of the k-major tiles are zero -- the staged core_1 (all threads) and
// this wave's core-0 rows (the slices' loads never touch them)
__device__ __forceinline__ void t2_zero_pads(float* Bt, float* As, int q0, int q1, int r1, int r1p, int ldk, int tid, int lane) {
  const int np = r1p - r1;
  if (np == 0) return;
  for (int e = tid; e < q1 * np; e += kT2Threads) Bt[(e / np) * ldk + r1 + e % np] = 0.f;
  for (int e = lane; e < q0 * np; e += kWave) As[(e / np) * ldk + r1 + e % np] = 0.f;
}

__global__ __launch_bounds__(kT2Threads) void t2_fwd_kernel(Dims d, Plan P, CorePtrs C, float* __restrict__ rows,
                                                           float* __restrict__ zout, long long nzero, T2Lds L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  zero_output(zout, nzero);
  const int4 cr = P.chunk_rec[blockIdx.x];
  const int s = cr.x, start = cr.y, len = cr.z;
  if (len == 0) return;
  const int tid = threadIdx.x, lane = lane_id(), w = tid / kWave;
  const int q0 = d.q[0], q1 = d.q[1], r1 = d.r[1], D = d.D, ldk = L.ldk;
  float* Bt = sm + L.oBt;                 // [q1][ldk]: core_1[s][k][b] at Bt[b][k]
  float* As = sm + L.oA + w * q0 * ldk;   // [q0][ldk]: this wave's lookup
  const float* B1 = C.c[1] + (size_t)s * d.slice[1];
  for (int e = tid; e < r1 * q1; e += kT2Threads) Bt[(e % q1) * ldk + e / q1] = B1[e];
  const int r1p = (r1 + 3) & ~3;
  const bool v4 = r1p == r1;  // (work-group-uniform) whole float4 rows: vector loads of core 0's slices
  t2_zero_pads(Bt, As, q0, q1, r1, r1p, ldk, tid, lane);
  __syncthreads();
  const int nA4 = q0 * r1 / 4;
  for (int j = w; j < len; j += kWaves) {
    const int4 rec = P.lrec[start + j];
    const float* A1 = C.c[0] + (size_t)rec.y * d.slice[0];
    if (v4) {
      const float4* A4 = (const float4*)A1;
      for (int e = lane; e < nA4; e += kWave) {  // (one coalesced round for q0 r1 <= 256)
        const float4 v = A4[e];
        const int a = (4 * e) / r1, k = (4 * e) % r1;
        *(float4*)(As + a * ldk + k) = v;
      }
    } else {
      for (int e = lane; e < q0 * r1; e += kWave) As[(e / r1) * ldk + e % r1] = A1[e];
    }
    // (wave-private region: LDS operations of a wave complete in order, no barrier)
    for (int o = lane; o < D; o += kWave) {
      const int a = o / q1, b = o - a * q1;
      const float4* ar = (const float4*)(As + a * ldk);
      const float4* br = (const float4*)(Bt + b * ldk);
      float acc = 0.f;
      for (int k4 = 0; k4 < r1p / 4; ++k4) {  // k ascending: the reference's order of additions (+ 0 * 0 on the padding)
        const float4 x = ar[k4], y = br[k4];
        acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
      }
      rows[(size_t)rec.x * D + o] = acc;
    }
  }
}

template <int NB>  // d core_1 outputs per lane: r1 q1 <= 64 NB
__global__ __launch_bounds__(kT2Threads) void t2_bwd_kernel(Dims d, Plan P, CorePtrs C, int B, const int64_t* __restrict__ rowidx,
                                                           const float* __restrict__ d_output, Partials PC, T2Lds L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  zero_hot_counters(PC);
  const int4 cr = P.chunk_rec[blockIdx.x];
  const int s = cr.x, start = cr.y, len = cr.z;
  if (len == 0) return;
  const int tid = threadIdx.x, lane = lane_id(), w = tid / kWave;
  const int q0 = d.q[0], q1 = d.q[1], r1 = d.r[1], D = d.D, ldk = L.ldk, ldg = D + 4;
  float* Bt = sm + L.oBt;
  float* As = sm + L.oA + w * q0 * ldk;
  float* Gs = sm + L.oG + w * ldg;        // [q0][q1] gradient row of the lookup's bag
  float* Rd = sm + L.oR;                  // [waves][r1 q1] the waves' d core_1 sums
  const float* B1 = C.c[1] + (size_t)s * d.slice[1];
  for (int e = tid; e < r1 * q1; e += kT2Threads) Bt[(e % q1) * ldk + e / q1] = B1[e];
  const int r1p = (r1 + 3) & ~3;
  const bool v4 = r1p == r1;
  t2_zero_pads(Bt, As, q0, q1, r1, r1p, ldk, tid, lane);
  __syncthreads();
  const bool has_row = P.hdr[3] != 0;
  const int nA4 = q0 * r1p / 4, n1 = r1 * q1, k4n = r1p / 4;
  float accB[NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) accB[u] = 0.f;
  for (int j = w; j < len; j += kWaves) {
    const int4 rec = P.lrec[start + j];
    const int n = rec.x;
    const long long row = has_row ? (long long)P.lrow[start + j] : rowidx[n];
    const int table = PC.tableidx ? (int)PC.tableidx[n] : s / d.p[1];
    const float sw = PC.psw ? PC.psw[n] : 1.f;
    const float* gsrc = d_output + ((size_t)table * B + row) * D;
    const float* A1 = C.c[0] + (size_t)rec.y * d.slice[0];
    if (v4) {
      const float4* A4 = (const float4*)A1;
      for (int e = lane; e < nA4; e += kWave) {
        const float4 v = A4[e];
        const int a = (4 * e) / r1, k = (4 * e) % r1;
        *(float4*)(As + a * ldk + k) = v;
      }
    } else {
      for (int e = lane; e < q0 * r1; e += kWave) As[(e / r1) * ldk + e % r1] = A1[e];
    }
    for (int e = lane; e < D; e += kWave) Gs[e] = gsrc[e] * sw;
    // d core_0[a][k .. k+3] = sum_b G[a][b] * core_1[k .. k+3][b]: the lookup's partial row (sorted order, Plan::ipos)
    float* o0 = PC.pc[0] + (size_t)P.ipos[0][n] * d.slice[0];
    for (int e = lane; e < nA4; e += kWave) {
      const int a = e / k4n, k = 4 * (e - a * k4n);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int b = 0; b < q1; ++b) {
        const float g = Gs[a * q1 + b];
        const float4 y = *(const float4*)(Bt + b * ldk + k);
        acc.x = fmaf(g, y.x, acc.x); acc.y = fmaf(g, y.y, acc.y); acc.z = fmaf(g, y.z, acc.z); acc.w = fmaf(g, y.w, acc.w);
      }
      if (v4) {
        ((float4*)o0)[e] = acc;
      } else {  // (rows of r1 floats: not 16-byte aligned)
        float* o = o0 + a * r1 + k;
        o[0] = acc.x;
        if (k + 1 < r1) o[1] = acc.y;
        if (k + 2 < r1) o[2] = acc.z;
        if (k + 3 < r1) o[3] = acc.w;
      }
    }
    // d core_1[k][b] += sum_a core_0[a][k] * G[a][b]: this wave's running sum over its lookups of the chunk
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int o = lane + u * kWave;
      if (o < n1) {
        const int k = o / q1, b = o - k * q1;
        float v = accB[u];
        for (int a = 0; a < q0; ++a) v = fmaf(As[a * ldk + k], Gs[a * q1 + b], v);
        accB[u] = v;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int o = lane + u * kWave;
    if (o < n1) Rd[w * n1 + o] = accB[u];
  }
  __syncthreads();
  float* pc1 = PC.pc[1] + (size_t)cr.w * d.slice[1];
  for (int e = tid; e < n1; e += kT2Threads) {
    float v = Rd[e];
    for (int ww = 1; ww < kWaves; ++ww) v += Rd[ww * n1 + e];
    pc1[e] = v;
  }
}

// ---- four cores through the three-core kernels (round 4) -------------------------------------------------------------------
// The reference contracts a lookup's cores left to right whatever their number (tt_embeddings_cuda.cu:754-918, 993-1054); for four
// cores that is x_0 [q0 q1 x r2] times core 2's [r2 x q2 r3] -- two thirds of the lookup's multiply-adds -- on a slice that is the
// lookup's own (8 KB at r = 32: the generic kernels stream it from L2 per lookup, 362 us forward / 1265 us backward at the
// benchmark's batch).  Matrix-chain order says otherwise: contract the LAST TWO cores of the lookup first,
//     M_n [r2 x q2 q3] = core_2[i_2] [r2 q2 x r3] * core_3[i_3] [r3 x q3]          (2 r2 q2 r3 q3 multiply-adds: 8 K at r = 32),
// and the lookup is a THREE-core lookup (core_0[i_0], core_1[i_1], M_n) with last factor q2 q3: the shape-specialised kernels
// take it as it is -- same plan (pivot = core 1), M_n read per lookup where they read core_2's slice (RealDims::c2n).  Backward:
// the three-core kernel leaves d M_n where it leaves d core_2's partial rows, and
//     d core_2[i_2] += d M_n * core_3[i_3]^T,      d core_3[i_3] += core_2[i_2]^T * d M_n
// are the per-lookup partial rows reduce_apply sums like any other core's.  Sums are re-associated, results agree with the
// left-to-right order to rounding (tested against the oracle at the default tolerance).  Taken when q2 q3 <= 32 (round 5; 16 until the q2 = 32 templates) and the
// three-core geometry has a specialised kernel (exact or padded); everything else stays on the generic kernels.
// Limits of the route: q3 <= 8 (the helpers' instantiations), a core-2 slice of at most kT4Slice floats (staged in LDS by the
// gradient kernel: r = 32 with q2 = 4, r = 64 with q2 = 2).
constexpr int kT4Slice = 8192;
constexpr int kT4Stage = 1536;  // floats of a lookup's d M row + core-3 slice the gradient kernel stages per step (round 5: q2 q3 up to 32)
static bool t4_merge_dims(const Dims& d, Dims* d3) {
  // (merged last factor q2 q3 up to 16 -- the reference's default four-core factorings of D = 128 / 256 are [2,4,4,4] / [4,4,4,4] --
  //  wherever a three-core template holds it: spec_match below; the helpers are instantiated for q3 <= 8)
  if (d.T != 4 || g_disable_spec || (long long)d.q[2] * d.q[3] > 32 || d.q[3] > 8 || (long long)d.r[2] * d.q[2] * d.r[3] > kT4Slice || d.r[3] > 128 ||
      (long long)d.r[2] * d.q[2] * d.q[3] + (long long)d.r[3] * d.q[3] > kT4Stage)
    return false;
  Dims e = d;
  e.T = 3;
  e.q[2] = d.q[2] * d.q[3];
  e.r[3] = 1;
  e.r[4] = 0;
  e.p[2] = 1; e.p[3] = 0; e.q[3] = 0;
  e.slice[2] = d.r[2] * e.q[2];
  e.slice[3] = 0;
  e.S[2] = 0; e.S[3] = 0;
  if (!spec_match(e)) return false;
  if (d3) *d3 = e;
  return true;
}
size_t t4_scratch_floats(const Dims& d) { return t4_merge_dims(d, nullptr) ? (size_t)d.r[2] * d.q[2] * d.q[3] : 0; }


// M[i][rq][x3] = sum_k core_2[sid_2][rq][k] * core_3[sid_3][k][x3] for the lookup at pivot position i (rq = (kk, x2) of r2 q2):
// one thread per (i, rq) row -- float4 loads of its r3 values when r3 % 4 == 0 --, k ascending.  Also leaves the lookup's
// {n, sid_2, sid_3, row of its core-3 partial} at its place in core 2's SORTED order (Plan::t4o): the backward's gradient
// kernel walks that order and reaches a lookup's operands in two dependent loads instead of four.
template <int Q3>
__global__ __launch_bounds__(256) void t4_merge_kernel(Plan P, const float* __restrict__ c2, const float* __restrict__ c3,
                                                       float* __restrict__ M, int r2q2, int r3) {
  const long long total = (long long)P.hdr[2] * r2q2;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e / r2q2), rq = (int)(e % r2q2);
    const int4 rec = P.lrec[i];
    if (rq == 0) P.t4o[P.ipos[2][rec.x]] = make_int4(rec.x, rec.z, rec.w, P.ipos[3][rec.x]);
    const float* a = c2 + ((size_t)rec.z * r2q2 + rq) * r3;
    const float* b = c3 + (size_t)rec.w * r3 * Q3;
    float acc[Q3];
#pragma unroll
    for (int x = 0; x < Q3; ++x) acc[x] = 0.f;
    if ((r3 & 3) == 0) {
#pragma unroll 4
      for (int k = 0; k < r3; k += 4) {
        const float4 av = *(const float4*)(a + k);
        float bv[4 * Q3];  // core_3[k .. k+3][0 .. Q3): 4 Q3 consecutive floats, the same for every row of the lookup
#pragma unroll
        for (int v = 0; v < Q3; ++v) *(float4*)(bv + 4 * v) = *(const float4*)(b + k * Q3 + 4 * v);
#pragma unroll
        for (int x = 0; x < Q3; ++x) {
          acc[x] = fmaf(av.x, bv[0 * Q3 + x], acc[x]);
          acc[x] = fmaf(av.y, bv[1 * Q3 + x], acc[x]);
          acc[x] = fmaf(av.z, bv[2 * Q3 + x], acc[x]);
          acc[x] = fmaf(av.w, bv[3 * Q3 + x], acc[x]);
        }
      }
    } else {
      for (int k = 0; k < r3; ++k) {
        const float av = a[k];
#pragma unroll
        for (int x = 0; x < Q3; ++x) acc[x] = fmaf(av, b[k * Q3 + x], acc[x]);
      }
    }
    float* o = M + ((size_t)rec.x * r2q2 + rq) * Q3;  // (M goes by LOOKUP: RealDims::c2n)
#pragma unroll
    for (int x = 0; x < Q3; ++x) o[x] = acc[x];
  }
}

// The same product in core 2's SORTED order (r3 <= 32, r2 q2 <= 256: what the benchmark shapes are).  The kernel above reads a
// lookup's 16 KB core-2 slice from L2 for every lookup (168 MB at 10k lookups: 33 us); lookups of one slice are consecutive in the
// sorted order, so a thread keeps ITS row of the slice in registers across the run and only the 256-byte core-3 slices -- staged
// for the whole segment at once -- and the result rows move.  Step 1 leaves {n, sid_2, sid_3, ipos_3} at the lookups' places in
// that order (Plan::t4o, also what the gradient kernel walks); step 2 takes 4 .. 32 positions per work-group (by batch size).  Same order of
// additions as above: bit-identical M.
// (round 5) the backward of a step needs the SAME M and order the forward of the step left in the plan -- the cores do not change
// between the two -- so the forward's merge marks them valid (hdr[kHdrT4Valid], with the cores' addresses) and the backward's
// launches leave at once when they find that mark: 21 us of the 210 us step at the benchmark's batch.  Cleared by a plan build and
// by t4_apply23_kernel when a fused optimizer writes cores 2 / 3.
// (round 6) ... and by EVERY fused write to cores 2 / 3, whichever plan's backward made it: with two forwards outstanding on one
// module (`loss = m(i1, o1).sum() + m(i2, o2).sum(); loss.backward()`) backward A rewrites the cores while plan B's mark would still
// say valid, and backward B would mix a stale M with the updated cores 0 / 1 (the reference recomputes its intermediates from the
// current cores in every backward).  g_t4_epoch is a device-resident counter of such writes (one per device, part of the code
// object: no allocation); the forward's merge stores its value beside the mark, t4_apply23_kernel increments it, and a plan's M
// is valid only while the two agree.  Conservative: a fused four-core step of ANY module invalidates every plan's M.
__device__ int g_t4_epoch = 0;
__device__ __forceinline__ bool t4_valid(const Plan& P, const float* c2, const float* c3) {
  const int* h = P.hdr + kHdrT4Valid;
  return h[0] == 1 && h[1] == (int)(uintptr_t)c2 && h[2] == (int)((uintptr_t)c2 >> 32) && h[3] == (int)(uintptr_t)c3 &&
         h[4] == (int)((uintptr_t)c3 >> 32) && h[5] == __hip_atomic_load(&g_t4_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void t4_order_kernel(Plan P, const float* c2, const float* c3, int reuse) {
  if (reuse && t4_valid(P, c2, c3)) return;  // (grid-uniform: nobody writes the mark while a reusing launch runs)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && !reuse) {  // the forward's launch: the stream orders every later reader behind this launch AND the merge after it
    int* h = P.hdr + kHdrT4Valid;
    h[1] = (int)(uintptr_t)c2; h[2] = (int)((uintptr_t)c2 >> 32); h[3] = (int)(uintptr_t)c3; h[4] = (int)((uintptr_t)c3 >> 32);
    h[5] = __hip_atomic_load(&g_t4_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h[0] = 1;
  }
  if (i >= P.hdr[2]) return;
  const int4 rec = P.lrec[i];
  P.t4o[P.ipos[2][rec.x]] = make_int4(rec.x, rec.z, rec.w, P.ipos[3][rec.x]);
}
constexpr int kT4MSeg = 32;  // positions per work-group at most; t4_mseg() picks by batch size
template <int Q3>
__global__ __launch_bounds__(256) void t4_merge_sorted_kernel(Plan P, const float* __restrict__ c2, const float* __restrict__ c3,
                                                              float* __restrict__ M, int r2q2, int r3, int seg, int reuse) {
  extern __shared__ __attribute__((aligned(16))) float bs[];  // [seg][r3 Q3]: the segment's core-3 slices
  __shared__ int4 recs[kT4MSeg];
  if (reuse && t4_valid(P, c2, c3)) return;  // (the forward of this step left M in the plan: t4_order_kernel)
  const int nnz = P.hdr[2];
  const int p0 = blockIdx.x * seg, cnt = min(seg, nnz - p0);
  if (cnt <= 0) return;
  const int tid = threadIdx.x, n3 = r3 * Q3;
  if (tid < cnt) recs[tid] = P.t4o[p0 + tid];
  __syncthreads();
  for (int e = tid; e < cnt * n3; e += blockDim.x) {
    const int j = e / n3;
    bs[e] = c3[(size_t)recs[j].z * n3 + (e - j * n3)];
  }
  __syncthreads();
  const int rq = tid;
  if (rq >= r2q2) return;
  float a[32];
  int cur = -1;
  for (int j = 0; j < cnt; ++j) {
    const int4 rc = recs[j];
    if (rc.y != cur) {  // (work-group-uniform) a new slice: this thread's row of it
      cur = rc.y;
      const float* ar = c2 + ((size_t)cur * r2q2 + rq) * r3;
      if ((r3 & 3) == 0) {
#pragma unroll
        for (int k = 0; k < 32; k += 4)
          if (k < r3) { const float4 t = *(const float4*)(ar + k); a[k] = t.x; a[k + 1] = t.y; a[k + 2] = t.z; a[k + 3] = t.w; }
      } else {
#pragma unroll
        for (int k = 0; k < 32; ++k)
          if (k < r3) a[k] = ar[k];
      }
    }
    const float* b = bs + j * n3;
    float acc[Q3];
#pragma unroll
    for (int x = 0; x < Q3; ++x) acc[x] = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k)
      if (k < r3) {  // (uniform; k ascending)
#pragma unroll
        for (int x = 0; x < Q3; ++x) acc[x] = fmaf(a[k], b[k * Q3 + x], acc[x]);
      }
    float* o = M + ((size_t)rc.x * r2q2 + rq) * Q3;
#pragma unroll
    for (int x = 0; x < Q3; ++x) o[x] = acc[x];
  }
}

// Gradients of cores 2 and 3 from d M (by lookup).  Work-group g takes positions [g SEG, (g + 1) SEG) of core 2's SORTED order
// (Plan::perm[2]: lookups of one slice are consecutive), stages the slice once per run, and per lookup n of the run
//   (a) adds  d M_n [rq][x3] * core_3[sid_3][k][x3]  to the run's sum of d core_2[slice][rq][k]            (registers),
//   (b) writes core_2[slice]^T d M_n = the lookup's partial row of core 3 at its place in core 3's order  (Plan::ipos[3]).
// A run's sum is stored as ONE partial row at the run's first position: a slice's partial rows are then at off[2][s] and at the
// multiples of SEG inside its range -- no list, and SEG times fewer 16 KB rows than one per lookup (168 MB at 10k lookups).
// The next lookup's d M row and core-3 slice are fetched into registers while the current one is multiplied.
constexpr int kT4Threads = 256;
constexpr int kT4Batch = 16;  // lookups staged per step of the gradient kernel: ALL their loads are in flight together
// Round 5: the kernel was 89 us of a 210 us step at the benchmark's batch for 0.3 GFLOP -- a chain of eight steps of four lookups
// per work-group, two work-group barriers per LOOKUP around a cross-thread reduction of (b).  Now a step stages sixteen lookups'
// d M rows and core-3 slices in one round of loads, (a) runs over them out of registers without a barrier, and (b) is computed
// for all lookups of the step at once, every thread reducing its outputs over rq by itself: three barriers per sixteen lookups.
// Q consecutive floats from LDS at a multiple of Q floats behind a 16-byte aligned base: one ds_read_b128 / b64 where Q allows
template <int Q>
__device__ __forceinline__ void lds_row(const float* p, float (&v)[Q]) {
  if constexpr (Q % 4 == 0) {
#pragma unroll
    for (int i = 0; i < Q; i += 4) { const float4 t = *(const float4*)(p + i); v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w; }
  } else if constexpr (Q % 2 == 0) {
#pragma unroll
    for (int i = 0; i < Q; i += 2) { const float2 t = *(const float2*)(p + i); v[i] = t.x; v[i + 1] = t.y; }
  } else {
#pragma unroll
    for (int i = 0; i < Q; ++i) v[i] = p[i];
  }
}
template <int Q3, int NO>    // NO: outputs of (a) per thread (n2 <= NO * 256)
__global__ __launch_bounds__(kT4Threads) void t4_grad23_kernel(Plan P, const float* __restrict__ c2, const float* __restrict__ c3,
                                                              const float* __restrict__ dM, float* __restrict__ pc2,
                                                              float* __restrict__ pc3, int r2q2, int r3, int SEG) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int4 recs[kT4Batch];
  const int n2 = r2q2 * r3, n3 = r3 * Q3, pm = r2q2 * Q3;
  float* c2s = sm;                       // [r2q2][r3]         the run's core-2 slice
  float* gms = c2s + (n2 + 3) / 4 * 4;   // [LB][r2q2][Q3]     d M of the step's lookups (16-byte aligned: lds_row)
  float* c3s = gms + kT4Batch * pm;      // [LB][r3][Q3]       their core-3 slices
  const int tid = threadIdx.x, nnz = P.hdr[2];
  const int beg = blockIdx.x * SEG, end = min(nnz, beg + SEG);
  if (beg >= end) return;
  // r3 divides the work-group (every benchmark shape): a thread's outputs of (a) all have k = tid % r3, its rq advance by
  // 256 / r3 -- the core-3 row is read ONCE per lookup, d M's rows by vector, broadcast over the lanes that share rq
#ifndef TTX_T4_AFAST
#define TTX_T4_AFAST 1
#endif
#ifndef TTX_T4_BNEW
#define TTX_T4_BNEW 1
#endif
  const bool kfix = TTX_T4_AFAST && (kT4Threads % r3) == 0;
  const int kk = tid % r3, rq0 = tid / r3, rqs = kT4Threads / r3;
  float acc[NO];
  int ork[NO];  // (rq << 16 | k) of output tid + 256 u: the division happens once, not per lookup
#pragma unroll
  for (int u = 0; u < NO; ++u) {
    const int o = tid + u * kT4Threads, rq = o / r3;
    ork[u] = (rq << 16) | (o - rq * r3);
    acc[u] = 0.f;
  }
  int run0 = beg, cur_sid = -1;
  for (int j0 = beg; j0 < end; j0 += kT4Batch) {
    const int nb = min(kT4Batch, end - j0);
    __syncthreads();  // (the previous step's records and staged rows are free)
    if (tid < nb) recs[tid] = P.t4o[j0 + tid];
    __syncthreads();
    for (int e = tid; e < nb * pm; e += kT4Threads) {
      const int b = e / pm;
      gms[e] = dM[(size_t)recs[b].x * pm + (e - b * pm)];
    }
    for (int e = tid; e < nb * n3; e += kT4Threads) {
      const int b = e / n3;
      c3s[e] = c3[(size_t)recs[b].z * n3 + (e - b * n3)];
    }
    int b0 = 0;
    while (b0 < nb) {  // the runs (lookups of one core-2 slice) inside the step: one, rarely two  (work-group-uniform)
      const int sid = recs[b0].y;
      int b1 = b0 + 1;
      while (b1 < nb && recs[b1].y == sid) ++b1;
      if (sid != cur_sid) {  // a new run: flush the previous sum, stage this slice
        if (cur_sid >= 0) {
#pragma unroll
          for (int u = 0; u < NO; ++u) {
            const int o = tid + u * kT4Threads;
            if (o < n2) pc2[(size_t)run0 * n2 + o] = acc[u];
            acc[u] = 0.f;
          }
          __syncthreads();  // (everyone is done with the previous slice)
        }
        for (int e = tid; e < n2; e += kT4Threads) c2s[e] = c2[(size_t)sid * n2 + e];
        cur_sid = sid;
        run0 = j0 + b0;
      }
      __syncthreads();  // (the step's rows -- and a new slice -- are staged)
      // (a) d core_2[rq][k] += sum_x3 dM[rq][x3] * core_3[k][x3], lookup after lookup
      for (int b = b0; b < b1; ++b) {
        const float* gm = gms + b * pm;
        const float* c3l = c3s + b * n3;
        if (kfix) {
          float cv[Q3];
          lds_row<Q3>(c3l + kk * Q3, cv);
#pragma unroll
          for (int u = 0; u < NO; ++u) {
            if (tid + u * kT4Threads < n2) {
              float g[Q3];
              lds_row<Q3>(gm + (rq0 + u * rqs) * Q3, g);
              float v = acc[u];
#pragma unroll
              for (int x = 0; x < Q3; ++x) v = fmaf(g[x], cv[x], v);
              acc[u] = v;
            }
          }
          continue;
        }
#pragma unroll
        for (int u = 0; u < NO; ++u) {
          const int o = tid + u * kT4Threads;
          if (o < n2) {
            const int rq = ork[u] >> 16, k = ork[u] & 0xffff;
            float v = acc[u];
#pragma unroll
            for (int x = 0; x < Q3; ++x) v = fmaf(gm[rq * Q3 + x], c3l[k * Q3 + x], v);
            acc[u] = v;
          }
        }
      }
      // (b) the lookups' rows of d core_3: [k][x3] = sum_rq core_2[rq][k] * dM[rq][x3], rq ascending.  Thread = (k, lookup group):
      // two lookups and all Q3 values of x3 per pass, so that one read of core_2[rq][k] feeds 2 Q3 multiply-adds and d M's rows
      // come by vector (three LDS reads per 2 Q3 multiply-adds; one read per multiply-add made the kernel LDS-bound)
      if (!TTX_T4_BNEW) {
        for (int o = tid; o < (b1 - b0) * n3; o += kT4Threads) {
          const int bl = o / n3, o3 = o - bl * n3, b = b0 + bl;
          const int k = o3 / Q3, x = o3 - k * Q3;
          const float* gm = gms + b * pm + x;
          const float* cc = c2s + k;
          float v = 0.f;
#pragma unroll 8
          for (int rq = 0; rq < r2q2; ++rq) v = fmaf(cc[rq * r3], gm[rq * Q3], v);
          pc3[(size_t)recs[b].w * n3 + o3] = v;
        }
      } else if (rq0 < rqs) {
        for (int bA = b0 + rq0; bA < b1; bA += 2 * rqs) {
          const int bB = bA + rqs;
          const bool hasB = bB < b1;
          const float* gA = gms + bA * pm;
          const float* gB = gms + (hasB ? bB : bA) * pm;
          const float* cc = c2s + kk;
          float vA[Q3], vB[Q3];
#pragma unroll
          for (int x = 0; x < Q3; ++x) { vA[x] = 0.f; vB[x] = 0.f; }
#pragma unroll 4
          for (int rq = 0; rq < r2q2; ++rq) {
            const float c = cc[rq * r3];
            float a[Q3], bq[Q3];
            lds_row<Q3>(gA + rq * Q3, a);
            lds_row<Q3>(gB + rq * Q3, bq);
#pragma unroll
            for (int x = 0; x < Q3; ++x) { vA[x] = fmaf(c, a[x], vA[x]); vB[x] = fmaf(c, bq[x], vB[x]); }
          }
          float* oA = pc3 + (size_t)recs[bA].w * n3 + kk * Q3;
#pragma unroll
          for (int x = 0; x < Q3; ++x) oA[x] = vA[x];
          if (hasB) {
            float* oB = pc3 + (size_t)recs[bB].w * n3 + kk * Q3;
#pragma unroll
            for (int x = 0; x < Q3; ++x) oB[x] = vB[x];
          }
        }
      }
      b0 = b1;
    }
  }
#pragma unroll
  for (int u = 0; u < NO; ++u) {
    const int o = tid + u * kT4Threads;
    if (o < n2) pc2[(size_t)run0 * n2 + o] = acc[u];
  }
}

// The same on the matrix pipe (round 5), for r2 q2 and r3 multiples of 16 -- the benchmark shapes.  Both products are small GEMMs
// once a step's sixteen lookups are stacked:
//   (a) d core_2[rq][k]  = sum over (lookup b, x3)  dM_b[rq][x3] * core_3_b[k][x3]     M = r2 q2, N = r3, K = lookups * q3
//   (b) d core_3_b[k][x3] = sum over rq             core_2[rq][k] * dM_b[rq][x3]       M = r3, N = lookups * q3, K = r2 q2
// v_mfma_f32_16x16x4_f32 (exact fp32): (a) keeps NA accumulator tiles per wave over the whole run, (b) one tile at a time.  The VALU
// form above spends ~4 k wave instructions of 4 cycles per step and work-group for 0.5 MFLOP: 40 of the 175 us step at the
// benchmark's batch, 850 us at 327k lookups.  Same staging, same partial-row contract; products added in another order (K in steps
// of four): equal to the VALU form to rounding.
// LDS layout of the matrix-pipe form: row strides chosen for conflict-free operand reads -- d M rows of a lookup pmS floats apart with
// pmS = 2 q3 (mod 32) (the 16 columns (lookup, x3) of a (b) tile x the two k rows of a half-wave fall on 32 different banks), core 2's
// rows r3S = 16 (mod 32) floats apart; both multiples of 4 (float4 staging).
struct T4Lds { int pmS, r3S, oG, oC, floats; };
static T4Lds t4_lds(int r2q2, int r3, int q3) {
  T4Lds L;
  const int pm = r2q2 * q3;
  L.pmS = (q3 == 2 || q3 == 4 || q3 == 8) ? pm + (((2 * q3 - pm) % 32) + 32) % 32 : (pm + 3) / 4 * 4;
  L.r3S = r3 + (((16 - r3) % 32) + 32) % 32;
  L.oG = r2q2 * L.r3S;
  L.oC = L.oG + kT4Batch * L.pmS;
  L.floats = L.oC + kT4Batch * r3 * q3;
  return L;
}
template <int Q3, int NA>  // NA: accumulator tiles of (a) per wave (r2 q2 r3 / 256 <= 4 NA)
__global__ __launch_bounds__(kT4Threads) void t4_grad23_mfma_kernel(Plan P, const float* __restrict__ c2, const float* __restrict__ c3,
                                                                   const float* __restrict__ dM, float* __restrict__ pc2,
                                                                   float* __restrict__ pc3, int r2q2, int r3, int SEG, T4Lds L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int4 recs[kT4Batch];
  const int n2 = r2q2 * r3, n3 = r3 * Q3, pm = r2q2 * Q3, pmS = L.pmS, r3S = L.r3S;
  float* c2s = sm;            // [r2q2][r3S]
  float* gms = sm + L.oG;     // [LB][pmS]     d M of the step's lookups, rows [rq][x3]
  float* c3s = sm + L.oC;     // [LB][r3][Q3]  their core-3 slices
  const int tid = threadIdx.x, nnz = P.hdr[2];
  const int lane = tid & 63, w = tid >> 6, i16 = lane & 15, kq = lane >> 4;
  const int beg = blockIdx.x * SEG, end = min(nnz, beg + SEG);
  if (beg >= end) return;
  const int NT = r3 / 16, ntiles = (r2q2 / 16) * NT;  // tiles of (a): t = mt NT + nt; wave w owns t = w NA .. w NA + NA - 1
  f32x4 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto flush = [&](int run0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int t = w * NA + i;
      if (t < ntiles) {
        const int mt = t / NT, nt = t - mt * NT;
        float* o = pc2 + (size_t)run0 * n2 + (size_t)(mt * 16 + kq * 4) * r3 + nt * 16 + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r * r3] = acc[i][r];
      }
      acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  int run0 = beg, cur_sid = -1;
  const int sb = tid >> 4, sl = tid & 15;  // staging: sixteen threads per lookup, float4 each
  // (measured and not kept: the NEXT step's records, d M rows and core-3 slices fetched into registers while the current step is
  //  multiplied -- 339 -> 409 us at 327k lookups, 20.2 -> 22.9 us at 10k: the steps are not latency-bound, the 40 extra registers cost
  //  more than the overlap gives)
  for (int j0 = beg; j0 < end; j0 += kT4Batch) {
    const int nb = min(kT4Batch, end - j0);
    __syncthreads();
    if (tid < nb) recs[tid] = P.t4o[j0 + tid];
    __syncthreads();
    if (sb < nb) {  // every load of the step in flight at once: the lookup's d M row and core-3 slice
      const float4* src = (const float4*)(dM + (size_t)recs[sb].x * pm);
      float4* dst = (float4*)(gms + sb * pmS);
      for (int e = sl; e < pm / 4; e += 16) dst[e] = src[e];
      const float4* s3 = (const float4*)(c3 + (size_t)recs[sb].z * n3);
      float4* d3 = (float4*)(c3s + sb * n3);
      for (int e = sl; e < n3 / 4; e += 16) d3[e] = s3[e];
    }
    int b0 = 0;
    while (b0 < nb) {  // the runs inside the step (work-group-uniform)
      const int sid = recs[b0].y;
      int b1 = b0 + 1;
      while (b1 < nb && recs[b1].y == sid) ++b1;
      if (sid != cur_sid) {
        if (cur_sid >= 0) {
          flush(run0);
          __syncthreads();
        }
        const float4* src = (const float4*)(c2 + (size_t)sid * n2);
        for (int e = tid; e < n2 / 4; e += kT4Threads) {  // (r3 % 16 == 0: a float4 stays inside one row)
          const int row = (e * 4) / r3, col = e * 4 - row * r3;
          *(float4*)(c2s + row * r3S + col) = src[e];
        }
        cur_sid = sid;
        run0 = j0 + b0;
      }
      __syncthreads();
      // (a): K = (b1 - b0) Q3 in steps of four; lane (kq, i16) feeds K index 4 ks + kq = (lookup, x3)
      const int K = (b1 - b0) * Q3;
      for (int ks = 0; ks < K; ks += 4) {
        const int kidx = ks + kq;
        const bool kv = kidx < K;
        const int b = b0 + (kv ? kidx / Q3 : 0), x = kv ? kidx % Q3 : 0;
        const float* ga = gms + b * pmS + x;
        const float* cb = c3s + b * n3 + x;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int t = w * NA + i;
          if (t < ntiles) {  // (wave-uniform)
            const int mt = t / NT, nt = t - mt * NT;
            const float av = kv ? ga[(mt * 16 + i16) * Q3] : 0.f;
            const float bv = kv ? cb[(nt * 16 + i16) * Q3] : 0.f;
            acc[i] = mfma4(av, bv, acc[i]);
          }
        }
      }
      // (b), transposed: rows = the run's (lookup, x3) pairs, columns = k, K = r2 q2 -- the accumulator's four registers are then
      // four consecutive (lookup, x3): with q3 = 4 one lookup's x3 = 0..3 (a float4 of its partial row), with q3 = 2 two lookups'
      // (a float2 each), and the sixteen lanes of a quarter write consecutive k: whole 64 .. 256-byte pieces of the rows instead of
      // scattered dwords (2.6 MB of 4-byte stores were most of this phase)
      const int Nb = (b1 - b0) * Q3, mtb = (Nb + 15) / 16, tb = mtb * NT;
      for (int t = w; t < tb; t += kT4Threads / 64) {
        const int mt = t / NT, nt = t - mt * NT;
        const int m = mt * 16 + i16;             // A operand: row (lookup, x3) = m
        const bool mv = m < Nb;
        const float* ga = gms + (b0 + (mv ? m / Q3 : 0)) * pmS + (mv ? m % Q3 : 0);
        const float* cb = c2s + nt * 16 + i16;   // B operand: column k
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int rq = kq; rq < r2q2; rq += 4) o = mfma4(mv ? ga[rq * Q3] : 0.f, cb[rq * r3S], o);
        const int m0 = mt * 16 + kq * 4, k = nt * 16 + i16;  // this lane's rows m0 .. m0 + 3, column k
        if constexpr (Q3 == 4) {
          if (m0 < Nb) *(float4*)(pc3 + (size_t)recs[b0 + m0 / 4].w * n3 + k * 4) = make_float4(o[0], o[1], o[2], o[3]);
        } else if constexpr (Q3 == 2) {
          if (m0 < Nb) *(float2*)(pc3 + (size_t)recs[b0 + m0 / 2].w * n3 + k * 2) = make_float2(o[0], o[1]);
          if (m0 + 2 < Nb) *(float2*)(pc3 + (size_t)recs[b0 + m0 / 2 + 1].w * n3 + k * 2) = make_float2(o[2], o[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (m0 + r < Nb) pc3[(size_t)recs[b0 + (m0 + r) / Q3].w * n3 + k * Q3 + (m0 + r) % Q3] = o[r];
        }
      }
      b0 = b1;
    }
  }
  flush(run0);
}

// ... and their reduction + optimizer.  Core 2: partial rows at off[2][s] and at the multiples of SEG inside the slice's range, in
// that order -- few, long rows: a slice is cut into blocks of 4 x 256 elements, one work-group each (round 5: one work-group per
// slice left 58 + 58 work-groups to do the step's 33 us at the benchmark's batch).  Core 3: one row per lookup, rows
// [off[3][s], off[3][s+1]) -- many short rows: one work-group per slice, G = 256 / V row groups, eight rows in flight per thread.
// DENSE writes the gradient (zeros for an untouched slice); SGD / Adagrad touch every element of every touched slice.
constexpr int kT4ApplyBlock = 4 * kT4Threads;
__global__ __launch_bounds__(kT4Threads) void t4_apply23_kernel(Plan P, const float* __restrict__ pc2, const float* __restrict__ pc3,
                                                               int S2, int n2, int n3, int SEG, int optim, float lr, float eps,
                                                               float* w2, float* w3, float* st2, float* st3, float* dw2,
                                                               float* dw3) {
  __shared__ float red[kT4Threads];
  if (blockIdx.x == 0 && threadIdx.x == 0 && optim != TTX_OPTIM_DENSE) {  // cores 2 / 3 change: this plan's M, and every other plan's, is stale
    P.hdr[kHdrT4Valid] = 0;
    atomicAdd(&g_t4_epoch, 1);
  }
  const int nb2 = (n2 + kT4ApplyBlock - 1) / kT4ApplyBlock;  // blocks of a core-2 slice
  const bool is2 = (int)blockIdx.x < S2 * nb2;
  const int s = is2 ? blockIdx.x / nb2 : blockIdx.x - S2 * nb2;
  const int* off = is2 ? P.off[2] : P.off[3];
  const int beg = off[s], end = off[s + 1], sl = is2 ? n2 : n3;
  const float* pc = is2 ? pc2 : pc3;
  float* w = (is2 ? w2 : w3) + (size_t)s * sl;
  float* st = (is2 ? st2 : st3);
  float* dw = (is2 ? dw2 : dw3);
  const int tid = threadIdx.x;
  auto emit = [&](int e, float g) {
    if (optim == TTX_OPTIM_DENSE) {
      dw[(size_t)s * sl + e] = g;
    } else {
      float sv = optim == TTX_OPTIM_ADAGRAD ? st[(size_t)s * sl + e] : 0.f;
      w[e] = apply_one(optim, g, w[e], lr, eps, &sv);
      if (optim == TTX_OPTIM_ADAGRAD) st[(size_t)s * sl + e] = sv;
    }
  };
  if (is2) {
    const int e0 = (blockIdx.x - s * nb2) * kT4ApplyBlock;
    if (beg >= end) {
      if (optim == TTX_OPTIM_DENSE)
        for (int e = e0 + tid; e < min(sl, e0 + kT4ApplyBlock); e += kT4Threads) dw[(size_t)s * sl + e] = 0.f;
      return;
    }
    const int r1 = (beg / SEG + 1) * SEG;  // rows: beg, then r1, r1 + SEG, .. below end; added in that order
    float g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + tid + u * kT4Threads;
      g[u] = e < sl ? pc[(size_t)beg * sl + e] : 0.f;
    }
    int r = r1;
    for (; r + 3 * SEG < end; r += 4 * SEG) {  // four rows x four elements in flight
      float x[4][4];
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + tid + u * kT4Threads;
          x[v][u] = e < sl ? pc[(size_t)(r + v * SEG) * sl + e] : 0.f;
        }
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] += x[v][u];
    }
    for (; r < end; r += SEG)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + tid + u * kT4Threads;
        if (e < sl) g[u] += pc[(size_t)r * sl + e];
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + tid + u * kT4Threads;
      if (e < sl) emit(e, g[u]);
    }
    return;
  }
  if (beg >= end) {
    if (optim == TTX_OPTIM_DENSE)
      for (int e = tid; e < sl; e += kT4Threads) dw[(size_t)s * sl + e] = 0.f;
    return;
  }
  // core 3: G = 256 / V row groups of V = min(sl, 256) lanes, group g sums rows beg + g, beg + g + G, .. with eight in flight; the
  // group sums are added in group order
  for (int e0 = 0; e0 < sl; e0 += kT4Threads) {
    const int V = min(sl - e0, kT4Threads), G = kT4Threads / V;
    const int g = tid / V, v = tid - g * V;
    float acc = 0.f;
    if (g < G) {
      int r = beg + g;
      for (; r + 7 * G < end; r += 8 * G) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = pc[(size_t)(r + u * G) * sl + e0 + v];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u];
      }
      for (; r < end; r += G) acc += pc[(size_t)r * sl + e0 + v];
    }
    __syncthreads();
    red[tid] = acc;
    __syncthreads();
    if (g == 0) {
      for (int k = 1; k < G; ++k) acc += red[k * V + v];
      emit(e0 + v, acc);
    }
  }
}
static int t4_grid(long long work) {
  const long long b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}
// positions of core 2's sorted order per work-group of the gradient kernel: few enough rows per slice at large batches, enough
// work-groups at small ones
static int t4_seg(long long nnz) {
  static const int forced = getenv("TTX_T4_SEG") ? atoi(getenv("TTX_T4_SEG")) : 0;  // (A/B)
  if (forced > 0) return forced;
  return nnz >= (1 << 18) ? 128 : (nnz >= (1 << 16) ? 64 : (nnz >= (1 << 14) ? 32 : 16));  // (a step of the kernel = 16 positions)
}
#define TTX_T4_Q3(Q3, CALL)                                  \
  switch (Q3) {                                              \
    case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
    case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; default: CALL(8); break; \
  }
static int t4_merge(const Dims& d, long long nnz, const Plan& P, const float* c2, const float* c3, hipStream_t st, int reuse) {
  const int r2q2 = d.r[2] * d.q[2];
  static const bool old_form = getenv("TTX_T4_OLD_MERGE") != nullptr;  // (A/B)
  if (d.r[3] <= 32 && r2q2 <= 256 && !old_form) {  // the sorted-order form (t4_merge_sorted_kernel)
    hipLaunchKernelGGL(t4_order_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, P, c2, c3, reuse);
    const int threads = (r2q2 + 63) / 64 * 64;
    // positions per work-group: the slice's rows are re-read once per segment, but a small batch needs the work-groups more than
    // the reuse (10k lookups, ms/step: 4 -> 0.221, 8 -> 0.225, 16 -> 0.244, 32 -> 0.273; the per-lookup form 0.247)
    const int seg = nnz >= (1 << 17) ? 32 : (nnz >= (1 << 15) ? 16 : 4);
    const size_t lds = (size_t)seg * d.r[3] * d.q[3] * sizeof(float);
#define TTX_T4_CALL(Q) hipLaunchKernelGGL(t4_merge_sorted_kernel<Q>, dim3((unsigned)((nnz + seg - 1) / seg)), dim3(threads), lds, st, P, c2, c3, P.t4m, r2q2, d.r[3], seg, reuse)
    TTX_T4_Q3(d.q[3], TTX_T4_CALL)
#undef TTX_T4_CALL
    TTX_HIP(hipGetLastError());
    return TTX_OK;
  }
#define TTX_T4_CALL(Q) hipLaunchKernelGGL(t4_merge_kernel<Q>, dim3(t4_grid(nnz * r2q2)), dim3(256), 0, st, P, c2, c3, P.t4m, r2q2, d.r[3])
  TTX_T4_Q3(d.q[3], TTX_T4_CALL)
#undef TTX_T4_CALL
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

// ---------------------------------------------------------- host side ------

// the generic kernels' carve for the block walk choose_tiles() picked (the plan was cut into chunks of its MC)
static int generic_lds(const Dims& d, const Plan& P, bool bwd, Lds* L) {
  const TileCfg cfg = choose_tiles(d);
  if (cfg.MC <= 0 || cfg.MC != P.MC)
    TTX_FAIL(TTX_EUNSUPPORTED,
             "no tiling of this TT shape fits %d B of LDS (one q1 block of x_0: %d x %d floats per lookup; ranks %d, %d)",
             g_lds_budget, d.q[0], d.T >= 3 ? d.k[1] : d.n[0], d.k[0], d.T >= 3 ? d.k[1] : 1);
  *L = make_lds(d, cfg.MC, bwd, cfg.bpp, cfg.KB);
  if (L->bytes > 160 * 1024) TTX_FAIL(TTX_EUNSUPPORTED, "internal: LDS carve of %d B", L->bytes);
  return TTX_OK;
}

template <typename K>
static int allow_lds(K kernel, int bytes) {
  if (bytes > 64 * 1024) return allow_dynamic_lds((const void*)kernel, bytes);
  return TTX_OK;
}

static size_t rows_bytes(const Dims& d, long long nnz) { return align_up((size_t)nnz * d.D * 4); }

static TTX_KNOB(int, g_skip_launch, 0);  // ablation (ttx_debug_skip bits 9..11): results invalid when != 0
static TTX_KNOB(int, g_no_pack, 0);      // A/B (ttx_debug_skip bit 16): reduce_apply with a work-group per small slice, as before round 6

// the family's translation unit takes it (ttx_tt_spec{16,32,64,128a,128b,128c}.hip)
static int run_rows_spec(TTX_SPEC_FWD_ARGS) {
  int rc = spec_fwd_32(id, P, C, rows, zout, nzero, F, fused, pad, R, st);
  if (rc == kSpecNotMine) rc = spec_fwd_64(id, P, C, rows, zout, nzero, F, fused, pad, R, st);
  if (rc == kSpecNotMine) rc = spec_fwd_16(id, P, C, rows, zout, nzero, F, fused, pad, R, st);
  if (rc == kSpecNotMine) rc = spec_fwd_128a(id, P, C, rows, zout, nzero, F, fused, pad, R, st);
  if (rc == kSpecNotMine) rc = spec_fwd_128b(id, P, C, rows, zout, nzero, F, fused, pad, R, st);
  if (rc == kSpecNotMine) rc = spec_fwd_128c(id, P, C, rows, zout, nzero, F, fused, pad, R, st);
  if (rc == kSpecNotMine) TTX_FAIL(TTX_EUNSUPPORTED, "no specialised forward kernel");
  return rc;
}

static int run_bwd_spec(TTX_SPEC_BWD_ARGS) {
  int rc = spec_bwd_32(id, d, P, C, B, rowidx, d_output, PC, pad, R, st);
  if (rc == kSpecNotMine) rc = spec_bwd_64(id, d, P, C, B, rowidx, d_output, PC, pad, R, st);
  if (rc == kSpecNotMine) rc = spec_bwd_16(id, d, P, C, B, rowidx, d_output, PC, pad, R, st);
  if (rc == kSpecNotMine) rc = spec_bwd_128a(id, d, P, C, B, rowidx, d_output, PC, pad, R, st);
  if (rc == kSpecNotMine) rc = spec_bwd_128b(id, d, P, C, B, rowidx, d_output, PC, pad, R, st);
  if (rc == kSpecNotMine) rc = spec_bwd_128c(id, d, P, C, B, rowidx, d_output, PC, pad, R, st);
  if (rc == kSpecNotMine) TTX_FAIL(TTX_EUNSUPPORTED, "no specialised backward kernel");
  return rc;
}

// fuse: NULL, or the arguments of pooling inside the contraction kernel; *fused says whether the kernel that ran did it
static int run_rows(const Dims& d, long long nnz, const Plan& P, const float* const* cores,
                    float* rows, float* zout, long long nzero, hipStream_t st, const PoolFuse* fuse = nullptr,
                    bool* fused = nullptr) {
  if (fused) *fused = false;
  bool pad = false;
  Dims d3;
  if (t4_merge_dims(d, &d3)) {  // four cores: the last two contracted per lookup, then the three-core kernel (see t4_merge_kernel)
    if (!P.t4m) TTX_FAIL(TTX_EINVAL, "internal: the plan carries no scratch for the four-core route");
    const SpecId id = spec_match(d3, &pad);
    ProfScope ps(TTX_PROF_FWD, st);
    const int rcm = t4_merge(d, nnz, P, cores[2], cores[3], st, 0);
    if (rcm) return rcm;
    CorePtrs C;
    for (int t = 0; t < TTX_MAX_CORES; ++t) C.c[t] = nullptr;
    C.c[0] = (float*)cores[0]; C.c[1] = (float*)cores[1]; C.c[2] = P.t4m;
    RealDims R = real_dims(d3);
    R.c2n = 1;
    bool did = false;  // (no fused pooling on this route)
    const PoolFuse none{};
    return run_rows_spec(id, P, C, rows, zout, nzero, none, &did, pad, R, st);
  }
  if (t2_shape(d)) {  // two cores: the dedicated kernels (t2_fwd_kernel)
    const T2Lds L = t2_lds(d, false);
    CorePtrs C;
    for (int t = 0; t < TTX_MAX_CORES; ++t) C.c[t] = t < d.T ? (float*)cores[t] : nullptr;
    ProfScope ps(TTX_PROF_FWD, st);
    hipLaunchKernelGGL(t2_fwd_kernel, dim3(P.max_chunks), dim3(kT2Threads), L.floats * sizeof(float), st, d, P, C, rows, zout,
                       nzero, L);
    TTX_HIP(hipGetLastError());
    return TTX_OK;
  }
  if (const SpecId id = spec_match(d, &pad)) {
    CorePtrs C;
    for (int t = 0; t < TTX_MAX_CORES; ++t) C.c[t] = t < d.T ? (float*)cores[t] : nullptr;
    ProfScope ps(TTX_PROF_FWD, st);
    bool did = fuse != nullptr && fuse->arrive != nullptr && !pad;
    const PoolFuse none{};
    const int rc = run_rows_spec(id, P, C, rows, zout, nzero, fuse ? *fuse : none, &did, pad, real_dims(d), st);
    if (fused) *fused = did;
    return rc;
  }
  Lds L;
  int rc = generic_lds(d, P, false, &L);
  if (rc) return rc;
  rc = allow_lds(fwd_kernel, L.bytes);
  if (rc) return rc;
  CorePtrs C;
  for (int t = 0; t < TTX_MAX_CORES; ++t) C.c[t] = t < d.T ? (float*)cores[t] : nullptr;
  ProfScope ps(TTX_PROF_FWD, st);
  hipLaunchKernelGGL(fwd_kernel, dim3(P.max_chunks), dim3(kThreads), L.bytes, st, d, P, C, rows, L, zout, nzero);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

}  // namespace ttx

using namespace ttx;

extern "C" {

#ifdef TTX_TEST_HOOKS  // ---- the knobs' setters: libttx_hooks.so only (include/ttx_test_hooks.h) ----
// debug: device buffer of 16 int64 stamps per work-group (NULL = off), scripts/phase_times.py
int ttx_debug_stamps(void* device_buffer) {
  g_stamps = (long long*)device_buffer;
  return TTX_OK;
}

// ablation knob for scripts/ablate.py: skip kernel phases (results become invalid)
int ttx_debug_skip(int32_t mask) {
  g_disable_spec = (mask & 256) ? 1 : 0;  // bit 8: force the generic kernels (A/B tests)
  g_disable_pad = (mask & 32768) ? 1 : 0;  // bit 15: shape-specialised kernels for exact shapes only (A/B tests)
  g_no_pack = (mask >> 16) & 1;           // bit 16: no wave-per-slice packing in reduce_apply (results stay valid)
  g_skip_launch = (mask >> 9) & 63;       // bits 9..11: leave out the pooling launch / reduce_apply / reduce_apply's pivot slices (upper bounds); bit 12: no fused pooling (A/B)
  mask &= 255;
  g_debug_skip = mask;
  return TTX_OK;
}

// test knob: LDS budget of the generic kernels' tile search (0 = the hardware's 160 KiB).  A small budget sends small
// shapes through the block walk (K blocks x column passes) that large ranks need.
int ttx_debug_lds_budget(int32_t bytes) {
  if (bytes < 0 || bytes > 160 * 1024) TTX_FAIL(TTX_EINVAL, "LDS budget %d out of range", bytes);
  g_lds_budget = bytes ? bytes : 160 * 1024;
  return TTX_OK;
}

int ttx_set_chunk(int32_t mc) {
  if (mc < 0 || mc > 64) TTX_FAIL(TTX_EINVAL, "chunk %d out of range 0..64", mc);
  g_chunk_override = mc;
  return TTX_OK;
}

// experiment (round 6): the backward of the benchmark shape at large batches on bwd32_kernel -- eight lookups per wave on
// v_mfma_f32_32x32x2, persistent work-groups -- with this many lookups per chunk (a multiple of 32); 0 = spec_bwd_kernel
int ttx_debug_bwd32(int32_t lookups_per_chunk) {
  if (lookups_per_chunk < 0 || lookups_per_chunk > 1024 || lookups_per_chunk % 32)
    TTX_FAIL(TTX_EINVAL, "lookups per chunk %d: a multiple of 32 in 0..1024", lookups_per_chunk);
  g_bwd32_mc = lookups_per_chunk;
  return TTX_OK;
}
#endif  // TTX_TEST_HOOKS

// test helper: how the generic kernels would walk this geometry's core_1 slice
// out = {lookups per chunk, q1 blocks per column pass, rows per K block, column passes, K blocks, LDS bytes (backward)};
// all zero when a shape-specialised kernel takes the geometry
int ttx_debug_tiles(const ttx_geom* g, int32_t* out) {
  Dims d;
  int rc = make_dims(g, &d);
  if (rc) return rc;
  if (!out) TTX_FAIL(TTX_EINVAL, "out is NULL");
  for (int i = 0; i < 6; ++i) out[i] = 0;
  if (spec_shape(d) || t4_merge_dims(d, nullptr) || t2_shape(d)) return TTX_OK;
  const TileCfg cfg = choose_tiles(d);
  if (cfg.MC <= 0) return TTX_OK;
  const Lds L = make_lds(d, cfg.MC, true, cfg.bpp, cfg.KB);
  out[0] = L.MC; out[1] = L.bpp; out[2] = L.KB; out[3] = L.ncp; out[4] = L.nkb; out[5] = L.bytes;
  return TTX_OK;
}

// Which of the TEST / ablation knobs are away from their defaults (0 = none): bit 0 ttx_debug_skip, 1 ttx_debug_lds_budget,
// 2 ttx_set_chunk, 3 ttx_debug_stamps, 5 ttx_debug_cache_fwd, 7 ttx_debug_bwd32.  The product build (libttx.so) has no knobs: this is the constant 0
// there, and bench.py refuses to time a library where it is not.  In libttx_hooks.so the knobs are plain globals -- not per
// stream, not thread-safe: tests and A/B timing only.
int ttx_debug_state(void) {
  return ((g_debug_skip | g_skip_launch | g_disable_spec | g_disable_pad | g_no_pack) ? 1 : 0) | (g_lds_budget != 160 * 1024 ? 2 : 0) |
         (g_chunk_override ? 4 : 0) | (g_stamps ? 8 : 0) | (ttx_cache_debug_state() << 4) | (g_bwd32_mc ? 128 : 0);
}

/* 1 = this library was built with -DTTX_TEST_HOOKS (libttx_hooks.so), 0 = the product build */
int ttx_has_test_hooks(void) {
#ifdef TTX_TEST_HOOKS
  return 1;
#else
  return 0;
#endif
}

size_t ttx_tt_forward_workspace_bytes(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz) {
  (void)B; (void)D;
  Dims d;
  if (make_dims(g, &d) != TTX_OK || nnz < 0) return 0;
  return plan_bytes(d, nnz) + rows_bytes(d, nnz) + 256;
}

static int common_checks(const Dims& d, int32_t D, int64_t nnz) {
  if (D <= 0) TTX_FAIL(TTX_EINVAL, "D=%d must be > 0", D);
  if (D != d.D) TTX_FAIL(TTX_EINVAL, "D=%d does not match prod(q)=%d", D, d.D);
  if (nnz < 0 || nnz >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "nnz=%lld out of range", (long long)nnz);
  if (choose_chunk(d, nnz) <= 0)
    TTX_FAIL(TTX_EUNSUPPORTED, "no tiling of this TT shape fits the LDS (one q1 block of x_0 is %d x %d floats per lookup)",
             d.q[0], d.T >= 3 ? d.k[1] : d.n[0]);
  return TTX_OK;
}

int ttx_tt_forward(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* indices,
                   const int64_t* rowidx, const int64_t* tableidx, const float* const* tt_cores,
                   float* output, const void* plan, void* workspace, size_t workspace_bytes,
                   ttx_stream_t stream) {
  return ttx_tt_forward_w(g, B, D, nnz, indices, rowidx, tableidx, nullptr, tt_cores, output, plan, workspace,
                          workspace_bytes, stream);
}

int ttx_tt_forward_w(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* indices,
                     const int64_t* rowidx, const int64_t* tableidx, const float* psw,
                     const float* const* tt_cores, float* output, const void* plan, void* workspace,
                     size_t workspace_bytes, ttx_stream_t stream) {
  return ttx_tt_forward_wr(g, B, D, nnz, indices, rowidx, tableidx, psw, tt_cores, output, nullptr, plan, workspace,
                           workspace_bytes, stream);
}

int ttx_tt_forward_wr(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* indices,
                      const int64_t* rowidx, const int64_t* tableidx, const float* psw,
                      const float* const* tt_cores, float* output, float* rows_keep, const void* plan, void* workspace,
                      size_t workspace_bytes, ttx_stream_t stream) {
  return ttx_tt_forward_o(g, B, D, nnz, indices, rowidx, tableidx, psw, tt_cores, output, rows_keep, nullptr, nullptr, plan,
                          workspace, workspace_bytes, stream);
}

int64_t ttx_tt_forward_arrive_ints(const ttx_geom* g, int64_t nnz) {
  Dims d;
  bool pad = false;
  if (make_dims(g, &d) != TTX_OK || nnz <= 0 || nnz > kPoolSpanMin || d.D % 4 != 0 || !spec_match(d, &pad) || pad) return 0;
  return nnz;
}

static int tt_forward_impl(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* indices,
                           const int64_t* rowidx, const int64_t* tableidx, const float* psw,
                           const float* const* tt_cores, float* output, float* rows_keep, const int64_t* offsets,
                           int32_t* arrive, const void* plan, void* workspace, size_t workspace_bytes, ttx_stream_t stream,
                           const int32_t* cache_loc, const float* cache_weight);

int ttx_tt_forward_o(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* indices,
                     const int64_t* rowidx, const int64_t* tableidx, const float* psw,
                     const float* const* tt_cores, float* output, float* rows_keep, const int64_t* offsets,
                     int32_t* arrive, const void* plan, void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  return tt_forward_impl(g, B, D, nnz, indices, rowidx, tableidx, psw, tt_cores, output, rows_keep, offsets, arrive, plan, workspace,
                         workspace_bytes, stream, nullptr, nullptr);
}

// 1 when ttx_tt_forward_cached pools the contraction's rows and gathers the cache's in one launch for this call; 0: the caller runs
// ttx_tt_forward + ttx_cache_forward_n (large batches, D % 4 != 0, more than one table)
int ttx_tt_forward_cached_supported(const ttx_geom* g, int32_t D, int64_t nnz) {
  static const bool off = getenv("TTX_NO_FUSED_CACHE_GATHER") != nullptr;  // (A/B)
  return !off && g && g->num_tables == 1 && nnz > 0 && nnz <= kPoolSpanMin && D % 4 == 0 && D / 4 <= 4 * kThreads;
}

int ttx_tt_forward_cached(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* indices,
                          const int64_t* rowidx, const int64_t* tableidx, const float* const* tt_cores,
                          const int32_t* cache_loc, const float* cache_weight, float* output, const void* plan,
                          void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  if (!ttx_tt_forward_cached_supported(g, D, nnz)) TTX_FAIL(TTX_EUNSUPPORTED, "ttx_tt_forward_cached: not for this call");
  if (!plan || !cache_loc || !cache_weight) TTX_FAIL(TTX_EINVAL, "ttx_tt_forward_cached needs the plan of the misses, loc and cache_weight");
  if ((((uintptr_t)cache_weight) | ((uintptr_t)output)) & 15) TTX_FAIL(TTX_EINVAL, "cache_weight / output must be 16-byte aligned");
  return tt_forward_impl(g, B, D, nnz, indices, rowidx, tableidx, nullptr, tt_cores, output, nullptr, nullptr, nullptr, plan, workspace,
                         workspace_bytes, stream, cache_loc, cache_weight);
}

static int tt_forward_impl(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* indices,
                           const int64_t* rowidx, const int64_t* tableidx, const float* psw,
                           const float* const* tt_cores, float* output, float* rows_keep, const int64_t* offsets,
                           int32_t* arrive, const void* plan, void* workspace, size_t workspace_bytes, ttx_stream_t stream,
                           const int32_t* cache_loc, const float* cache_weight) {
  Dims d;
  int rc = make_dims(g, &d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (B < 0 || !output) TTX_FAIL(TTX_EINVAL, "bad B/output");
  const long long nout = (long long)d.num_tables * B * d.D;
  if (nnz == 0) {  // zeros, like cu:981-985
    if (nout > 0) TTX_HIP(hipMemsetAsync(output, 0, (size_t)nout * sizeof(float), st));
    return TTX_OK;
  }
  rc = common_checks(d, D, nnz);
  if (rc) return rc;
  if (!indices || !rowidx || !tableidx || !tt_cores) TTX_FAIL(TTX_EINVAL, "NULL input");
  const size_t pb = plan ? 0 : plan_bytes(d, nnz);
  if (!workspace || workspace_bytes < pb + rows_bytes(d, nnz))
    TTX_FAIL(TTX_EWORKSPACE, "forward workspace too small: %zu < %zu", workspace_bytes, pb + rows_bytes(d, nnz));
  char* ws = (char*)workspace;
  Plan P;
  if (plan) {
    P = carve_plan(d, nnz, (void*)plan);
  } else {
    P = carve_plan(d, nnz, ws);
    rc = plan_build(d, nnz, indices, tableidx, rowidx, P, st);
    if (rc) return rc;
    ws += pb;
  }
  if (rows_keep && (((uintptr_t)rows_keep) & 15)) TTX_FAIL(TTX_EINVAL, "rows_keep must be 16-byte aligned");
  float* rows = rows_keep ? rows_keep : (float*)ws;
  // pooling inside the contraction kernel (spec_fwd_kernel<.., FUSE>): the caller names the bags (offsets) and lends a
  // zeroed counter per lookup; small batches of the specialised shapes only -- everything else pools in a launch of its own
  PoolFuse F{};
  const bool offer = offsets && arrive && nnz <= kPoolSpanMin && d.D % 4 == 0 && (((uintptr_t)rows | (uintptr_t)output) & 15) == 0 &&
                     (size_t)nnz * d.D * 4 < (1ull << 31) && (long long)d.num_tables * B < (1ll << 30) && !(g_skip_launch & 8);
  if (offer) {
    F.arrive = arrive; F.offsets = offsets; F.rowidx = rowidx; F.tableidx = d.tab ? tableidx : nullptr; F.psw = psw;
    F.out = output; F.B = B; F.bags = d.num_tables * B; F.p1 = d.p[1]; F.rows_bytes = (unsigned)((size_t)nnz * d.D * 4);
    F.dbg = (g_skip_launch >> 4) & 3;
  }
  bool fused = false;
  rc = run_rows(d, nnz, P, tt_cores, rows, output, nout, st, offer ? &F : nullptr, &fused);  // also zeroes `output`
  if (rc) return rc;
  if (cache_loc) {  // (ttx_tt_forward_cached: both parts of the batch in one launch)
    ProfScope ps(TTX_PROF_POOL, st);
    hipLaunchKernelGGL(pool4_small_cached_kernel, dim3(((int)nnz + kThreads / 16 - 1) / (kThreads / 16)), dim3(kThreads), 0, st,
                       P.hdr, rowidx, (const float4*)rows, cache_loc, (const float4*)cache_weight, output, (int)nnz, d.D / 4);
    TTX_HIP(hipGetLastError());
  } else if (!(g_skip_launch & 1) && !fused) {
    ProfScope ps(TTX_PROF_POOL, st);
    if (d.D % 4 == 0 && (((uintptr_t)rows | (uintptr_t)output) & 15) == 0) {
      if (nnz <= kPoolSpanMin && (unsigned long long)nnz * d.D * 4 < (1ull << 32) - 16)
        hipLaunchKernelGGL(pool4_small_kernel, dim3(((int)nnz + kThreads / 16 - 1) / (kThreads / 16)), dim3(kThreads), 0, st,
                           P.hdr, rowidx, tableidx, (const float4*)rows, (float4*)output, (int)nnz, B, d.D / 4, psw);
      else
        hipLaunchKernelGGL(pool4_kernel, dim3(((int)nnz + kThreads - 1) / kThreads), dim3(kThreads), 0, st,
                           (int)nnz, P.hdr, B, d.D / 4, rowidx, tableidx, (const float4*)rows, psw, (float4*)output);
    } else {
      const int groups = kThreads / 32;
      hipLaunchKernelGGL(pool_kernel, dim3(((int)nnz + groups - 1) / groups), dim3(kThreads), 0, st,
                         (int)nnz, P.hdr, B, d.D, rowidx, tableidx, rows, psw, output);
    }
    TTX_HIP(hipGetLastError());
  }
  return TTX_OK;
}

int ttx_tt_rows(const ttx_geom* g, int32_t D, int64_t nnz, const int64_t* indices,
                const int64_t* tableidx, const float* const* tt_cores, float* rows,
                void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  Dims d;
  int rc = make_dims(g, &d);
  if (rc) return rc;
  if (nnz == 0) return TTX_OK;
  rc = common_checks(d, D, nnz);
  if (rc) return rc;
  if (!indices || !tt_cores || !rows) TTX_FAIL(TTX_EINVAL, "NULL input");
  if (!workspace || workspace_bytes < plan_bytes(d, nnz))
    TTX_FAIL(TTX_EWORKSPACE, "rows workspace too small: %zu < %zu", workspace_bytes, plan_bytes(d, nnz));
  Plan P = carve_plan(d, nnz, workspace);
  rc = plan_build(d, nnz, indices, tableidx, nullptr, P, (hipStream_t)stream);
  if (rc) return rc;
  return run_rows(d, nnz, P, tt_cores, rows, nullptr, 0, (hipStream_t)stream);
}

static int num_segments(const Dims& d, long long nnz, int MC, int t) {
  if (t == 1) return 0;  // (the pivot's hot slices are split by columns: hot_wgs)
  const long long total = (t == 1) ? (long long)max_chunks(d, nnz, MC) : nnz;
  const int SEG = (t == 1) ? kSegPivot : kSegThin;
  return (int)((total + SEG - 1) / SEG);
}

// backward scratch: partial gradients per core, then the hot-slice segment sums (2 slots per segment) and
// the arrival counters (one int per core slice).  offs[t] = partials of core t, offs[T + t] = segment sums
// of core t, offs[2T] = counters.
static size_t partial_bytes(const Dims& d, long long nnz, int MC, size_t* offs) {
  size_t o = 0;
  for (int t = 0; t < d.T; ++t) {
    offs[t] = o;
    const size_t cnt = (t == 1) ? (size_t)max_chunks(d, nnz, MC) : (size_t)nnz;
    o += align_up(cnt * d.slice[t] * sizeof(float));
  }
  for (int t = 0; t < d.T; ++t) {
    offs[d.T + t] = o;
    o += align_up((size_t)2 * num_segments(d, nnz, MC, t) * d.slice[t] * sizeof(float));
  }
  offs[2 * d.T] = o;
  size_t ns = 0;
  for (int t = 0; t < d.T; ++t) ns += d.S[t];
  o += align_up(ns * sizeof(int));
  return o;
}

size_t ttx_tt_backward_workspace_bytes(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz) {
  (void)B; (void)D;
  Dims d;
  if (make_dims(g, &d) != TTX_OK || nnz < 0) return 0;
  const int MC = choose_chunk(d, nnz);
  if (MC <= 0) return 0;
  size_t offs[2 * TTX_MAX_CORES + 1];
  return plan_bytes(d, nnz) + partial_bytes(d, nnz, MC, offs) + 256;
}

int ttx_tt_backward(const ttx_geom* g, int32_t optim, int32_t B, int32_t D, float lr, float eps,
                    int64_t nnz, const int64_t* indices, const int64_t* rowidx,
                    const int64_t* tableidx, const float* d_output, float* const* tt_cores,
                    float* const* optimizer_state, float* const* d_tt_cores, const void* plan,
                    void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  return ttx_tt_backward_w(g, optim, B, D, lr, eps, nnz, indices, rowidx, tableidx, nullptr, d_output, tt_cores,
                           optimizer_state, d_tt_cores, plan, workspace, workspace_bytes, stream);
}

static int tt_backward_impl(const ttx_geom* g, int32_t optim, int32_t B, int32_t D, float lr, float eps,
                            int64_t nnz, const int64_t* indices, const int64_t* rowidx,
                            const int64_t* tableidx, const float* psw, const float* d_output,
                            float* const* tt_cores, float* const* optimizer_state, float* const* d_tt_cores,
                            const void* plan, void* workspace, size_t workspace_bytes, ttx_stream_t stream,
                            const CacheTail* tail, int32_t* tail_done);

int ttx_tt_backward_w(const ttx_geom* g, int32_t optim, int32_t B, int32_t D, float lr, float eps,
                      int64_t nnz, const int64_t* indices, const int64_t* rowidx,
                      const int64_t* tableidx, const float* psw, const float* d_output,
                      float* const* tt_cores, float* const* optimizer_state, float* const* d_tt_cores,
                      const void* plan, void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  return tt_backward_impl(g, optim, B, D, lr, eps, nnz, indices, rowidx, tableidx, psw, d_output, tt_cores, optimizer_state,
                          d_tt_cores, plan, workspace, workspace_bytes, stream, nullptr, nullptr);
}

int ttx_tt_backward_wc(const ttx_geom* g, int32_t optim, int32_t B, int32_t D, float lr, float eps,
                       int64_t nnz, const int64_t* indices, const int64_t* rowidx,
                       const int64_t* tableidx, const float* psw, const float* d_output,
                       float* const* tt_cores, float* const* optimizer_state, float* const* d_tt_cores,
                       const void* plan, void* workspace, size_t workspace_bytes, ttx_stream_t stream,
                       const int32_t* skip_dev, const int32_t* cache_loc, const float* cache_grad, float cache_scale,
                       float* cache_dst, int32_t* tail_done) {
  static const bool off = getenv("TTX_NO_FUSED_CACHE_SCATTER") != nullptr;  // (A/B)
  if (!tail_done) TTX_FAIL(TTX_EINVAL, "tail_done is NULL");
  *tail_done = 0;
  CacheTail CT{};
  if (!off && cache_dst && cache_loc && cache_grad && nnz > 0 && nnz < (1ll << 31)) {
    CT.N = (int)nnz; CT.D = D; CT.scale = cache_scale; CT.skip_dev = skip_dev; CT.grad = cache_grad; CT.loc = cache_loc;
    CT.rowidx = rowidx; CT.dst = cache_dst;
  }
  return tt_backward_impl(g, optim, B, D, lr, eps, nnz, indices, rowidx, tableidx, psw, d_output, tt_cores, optimizer_state,
                          d_tt_cores, plan, workspace, workspace_bytes, stream, &CT, tail_done);
}

static int tt_backward_impl(const ttx_geom* g, int32_t optim, int32_t B, int32_t D, float lr, float eps,
                            int64_t nnz, const int64_t* indices, const int64_t* rowidx,
                            const int64_t* tableidx, const float* psw, const float* d_output,
                            float* const* tt_cores, float* const* optimizer_state, float* const* d_tt_cores,
                            const void* plan, void* workspace, size_t workspace_bytes, ttx_stream_t stream,
                            const CacheTail* tail, int32_t* tail_done) {
  Dims d;
  int rc = make_dims(g, &d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (optim != TTX_OPTIM_SGD && optim != TTX_OPTIM_ADAGRAD && optim != TTX_OPTIM_DENSE)
    TTX_FAIL(TTX_EINVAL, "unknown optimizer selector %d", optim);
  if (optim == TTX_OPTIM_DENSE && !d_tt_cores) TTX_FAIL(TTX_EINVAL, "d_tt_cores is NULL");
  if (optim == TTX_OPTIM_ADAGRAD && !optimizer_state) TTX_FAIL(TTX_EINVAL, "optimizer_state is NULL");
  if (nnz == 0) {
    if (optim == TTX_OPTIM_DENSE)  // zeros_like, cu:444-450
      for (int t = 0; t < d.T; ++t)
        TTX_HIP(hipMemsetAsync(d_tt_cores[t], 0, (size_t)d.S[t] * d.slice[t] * sizeof(float), st));
    return TTX_OK;
  }
  rc = common_checks(d, D, nnz);
  if (rc) return rc;
  if (!indices || !rowidx || !tableidx || !d_output || !tt_cores) TTX_FAIL(TTX_EINVAL, "NULL input");
  const int MC = choose_chunk(d, nnz);
  size_t offs[2 * TTX_MAX_CORES + 1];
  const size_t pcb = partial_bytes(d, nnz, MC, offs);
  const size_t pb = plan ? 0 : plan_bytes(d, nnz);
  if (!workspace || workspace_bytes < pb + pcb)
    TTX_FAIL(TTX_EWORKSPACE, "backward workspace too small: %zu < %zu", workspace_bytes, pb + pcb);
  char* ws = (char*)workspace;
  Plan P;
  if (plan) {
    P = carve_plan(d, nnz, (void*)plan);
  } else {
    P = carve_plan(d, nnz, ws);
    rc = plan_build(d, nnz, indices, tableidx, rowidx, P, st);
    if (rc) return rc;
    ws += pb;
  }
  Partials PC;
  for (int t = 0; t < TTX_MAX_CORES; ++t) PC.pc[t] = t < d.T ? (float*)(ws + offs[t]) : nullptr;
  PC.psw = psw;
  PC.tableidx = d.tab ? tableidx : nullptr;
  int nslices = 0, nsegs = 0;
  for (int t = 0; t < d.T; ++t) {
    PC.seg[t] = (float*)(ws + offs[d.T + t]);
    nslices += d.S[t];
    nsegs += (t == 1) ? hot_wgs(P.max_chunks, d.slice[1], 1) : num_segments(d, nnz, MC, t);
  }
  for (int t = d.T; t < TTX_MAX_CORES; ++t) PC.seg[t] = nullptr;
  PC.hot_cnt = (int*)(ws + offs[2 * d.T]);
  PC.n_hot_cnt = nslices;
  CorePtrs C, S, DW;
  for (int t = 0; t < TTX_MAX_CORES; ++t) {
    C.c[t] = t < d.T ? tt_cores[t] : nullptr;
    S.c[t] = (t < d.T && optim == TTX_OPTIM_ADAGRAD) ? optimizer_state[t] : nullptr;
    DW.c[t] = (t < d.T && optim == TTX_OPTIM_DENSE) ? d_tt_cores[t] : nullptr;
  }
  bool pad = false;
  bool t4_route = false;
  Dims d3;
  if (t4_merge_dims(d, &d3)) {  // four cores on the three-core kernel: M, the backward with d M where core 2's partials go, d M -> cores 2, 3
    if (!P.t4m || !P.t4g) TTX_FAIL(TTX_EINVAL, "internal: the plan carries no scratch for the four-core route");
    const SpecId id = spec_match(d3, &pad);
    const int r2q2 = d.r[2] * d.q[2], n2 = r2q2 * d.r[3], n3 = d.r[3] * d.q[3], pm = r2q2 * d.q[3];
    ProfScope ps(TTX_PROF_BWD, st);
    rc = t4_merge(d, nnz, P, tt_cores[2], tt_cores[3], st, 1);  // (reuses the forward's M when the plan still holds it)
    if (rc) return rc;
    CorePtrs C3;
    for (int t = 0; t < TTX_MAX_CORES; ++t) C3.c[t] = nullptr;
    C3.c[0] = tt_cores[0]; C3.c[1] = tt_cores[1]; C3.c[2] = P.t4m;
    Partials PC3 = PC;
    PC3.pc[2] = P.t4g;
    PC3.pc[3] = nullptr;
    RealDims R = real_dims(d3);
    R.c2n = 1;
    rc = run_bwd_spec(id, d3, P, C3, B, rowidx, d_output, PC3, pad, R, st);
    if (rc) return rc;
    const int SEG = t4_seg(nnz);
    const size_t lds = (size_t)((n2 + 3) / 4 * 4 + kT4Batch * (pm + n3)) * sizeof(float);
    const int gblocks = (int)((nnz + SEG - 1) / SEG);
    static const bool t4_valu = getenv("TTX_T4_VALU") != nullptr;  // (A/B: the VALU form for every shape)
    const bool t4_mfma = !t4_valu && r2q2 % 16 == 0 && d.r[3] % 16 == 0 && n2 / 256 <= 32;
    const T4Lds TL = t4_lds(r2q2, d.r[3], d.q[3]);
    const size_t lds_m = (size_t)TL.floats * sizeof(float);
#define TTX_T4_MFMA(Q)                                                                                                          \
    do {                                                                                                                         \
      if (n2 / 256 <= 16) {                                                                                                       \
        if (lds_m > 64 * 1024) { rc = allow_lds(t4_grad23_mfma_kernel<Q, 4>, (int)lds_m); if (rc) return rc; }                    \
        hipLaunchKernelGGL((t4_grad23_mfma_kernel<Q, 4>), dim3(gblocks), dim3(kT4Threads), lds_m, st, P, tt_cores[2], tt_cores[3],\
                           P.t4g, PC.pc[2], PC.pc[3], r2q2, d.r[3], SEG, TL);                                                      \
      } else {                                                                                                                   \
        if (lds_m > 64 * 1024) { rc = allow_lds(t4_grad23_mfma_kernel<Q, 8>, (int)lds_m); if (rc) return rc; }                    \
        hipLaunchKernelGGL((t4_grad23_mfma_kernel<Q, 8>), dim3(gblocks), dim3(kT4Threads), lds_m, st, P, tt_cores[2], tt_cores[3],\
                           P.t4g, PC.pc[2], PC.pc[3], r2q2, d.r[3], SEG, TL);                                                      \
      }                                                                                                                          \
    } while (0)
#define TTX_T4_CALL(Q)                                                                                                          \
    do {                                                                                                                         \
      if (t4_mfma) { TTX_T4_MFMA(Q); break; }                                                                                     \
      if (lds > 64 * 1024) {                                                                                                      \
        rc = n2 <= 8 * kT4Threads ? allow_lds(t4_grad23_kernel<Q, 8>, (int)lds)                                                   \
                                  : (n2 <= 16 * kT4Threads ? allow_lds(t4_grad23_kernel<Q, 16>, (int)lds) : allow_lds(t4_grad23_kernel<Q, 32>, (int)lds)); \
        if (rc) return rc;                                                                                                        \
      }                                                                                                                          \
      if (n2 <= 8 * kT4Threads)                                                                                                   \
        hipLaunchKernelGGL((t4_grad23_kernel<Q, 8>), dim3(gblocks), dim3(kT4Threads), lds, st, P, tt_cores[2], tt_cores[3], P.t4g, \
                           PC.pc[2], PC.pc[3], r2q2, d.r[3], SEG);                                                               \
      else if (n2 <= 16 * kT4Threads)                                                                                             \
        hipLaunchKernelGGL((t4_grad23_kernel<Q, 16>), dim3(gblocks), dim3(kT4Threads), lds, st, P, tt_cores[2], tt_cores[3], P.t4g, \
                           PC.pc[2], PC.pc[3], r2q2, d.r[3], SEG);                                                               \
      else                                                                                                                        \
        hipLaunchKernelGGL((t4_grad23_kernel<Q, 32>), dim3(gblocks), dim3(kT4Threads), lds, st, P, tt_cores[2], tt_cores[3], P.t4g, \
                           PC.pc[2], PC.pc[3], r2q2, d.r[3], SEG);                                                               \
    } while (0)
    TTX_T4_Q3(d.q[3], TTX_T4_CALL)
#undef TTX_T4_CALL
#undef TTX_T4_MFMA
    TTX_HIP(hipGetLastError());
    t4_route = true;
  } else if (t2_shape(d)) {
    const T2Lds L = t2_lds(d, true);
    const int n1 = d.r[1] * d.q[1];
    ProfScope ps(TTX_PROF_BWD, st);
#define TTX_T2_BWD(NB)                                                                                                        \
    do {                                                                                                                       \
      rc = allow_lds(t2_bwd_kernel<NB>, L.floats * (int)sizeof(float));                                                        \
      if (rc) return rc;                                                                                                       \
      hipLaunchKernelGGL(t2_bwd_kernel<NB>, dim3(P.max_chunks), dim3(kT2Threads), L.floats * sizeof(float), st, d, P, C, B,   \
                         rowidx, d_output, PC, L);                                                                             \
    } while (0)
    if (n1 <= 4 * kWave) TTX_T2_BWD(4);
    else if (n1 <= 8 * kWave) TTX_T2_BWD(8);
    else if (n1 <= 16 * kWave) TTX_T2_BWD(16);
    else TTX_T2_BWD(32);
#undef TTX_T2_BWD
    TTX_HIP(hipGetLastError());
  } else if (const SpecId id = spec_match(d, &pad)) {
    ProfScope ps(TTX_PROF_BWD, st);
    rc = run_bwd_spec(id, d, P, C, B, rowidx, d_output, PC, pad, real_dims(d), st);
    if (rc) return rc;
  } else {
    Lds L;
    rc = generic_lds(d, P, true, &L);
    if (rc) return rc;
    rc = allow_lds(bwd_kernel, L.bytes);
    if (rc) return rc;
    ProfScope ps(TTX_PROF_BWD, st);
    hipLaunchKernelGGL(bwd_kernel, dim3(P.max_chunks), dim3(kThreads), L.bytes, st, d, P, C, B,
                       rowidx, d_output, PC, L);
    TTX_HIP(hipGetLastError());
  }
  if (t4_route && !(g_skip_launch & 2)) {
    // four-core route: cores 0 and 1 through reduce_apply as a two-core geometry (the pivot is core 1 either way), cores 2 and 3
    // from the run sums / per-lookup rows the gradient kernel left (t4_apply23_kernel)
    Dims d2 = d;
    d2.T = 2;
    int ns2 = d.S[0] + d.S[1], nsg2 = num_segments(d, nnz, MC, 0) + hot_wgs(P.max_chunks, d.slice[1], 1);
    const int smax = d.slice[0] > d.slice[1] ? d.slice[0] : d.slice[1];
    const int rthreads = smax <= 4096 ? TTX_RTHREADS_SMALL : kReduceThreads;
    ProfScope ps(TTX_PROF_APPLY, st);
    hipLaunchKernelGGL(reduce_apply_kernel, dim3(ns2 + nsg2), dim3(rthreads), reduce_lds_bytes(rthreads), st, d2, P, PC, optim, lr, eps, C, S, DW, ns2,
                       (int)nnz, CacheTail{}, 1);
    TTX_HIP(hipGetLastError());
    hipLaunchKernelGGL(t4_apply23_kernel, dim3(d.S[2] * ((d.slice[2] + kT4ApplyBlock - 1) / kT4ApplyBlock) + d.S[3]), dim3(kT4Threads), 0, st, P, PC.pc[2], PC.pc[3], d.S[2],
                       d.slice[2], d.slice[3], t4_seg(nnz), optim, lr, eps, C.c[2], C.c[3], S.c[2], S.c[3], DW.c[2], DW.c[3]);
    TTX_HIP(hipGetLastError());
    return TTX_OK;
  }
  if (!(g_skip_launch & 2)) {
    int smax = 0;
    for (int t = 0; t < d.T; ++t) smax = d.slice[t] > smax ? d.slice[t] : smax;
    int rthreads = smax <= 4096 ? TTX_RTHREADS_SMALL : kReduceThreads;  // (measured: 512 is 1.7 us faster at r = 32, 1024 at r = 64)
    // (round 5) small slices, many of them -- two cores over 11 M rows are 2 x 3317 slices of 256 floats: 6634 work-groups of 512
    // threads for 64 float4 lanes of work each were 6.5 rounds of the chip's wave slots, 30 us -- get work-groups of their size
    if (smax <= 512) rthreads = 128;
    else if (smax <= 2048) rthreads = 256;
    // (round 6) every slice at most 64 float4 lanes (two cores; three cores at tiny ranks): a wave per slice, four per work-group
    int pack = 1, smin4 = 0;
    for (int t = 0; t < d.T; ++t) smin4 |= d.slice[t] & 3;
    if (smax <= 4 * kWave && smin4 == 0 && nslices >= 1024 && !g_no_pack) { pack = 4; rthreads = pack * kWave; }
    if (tail && tail->dst && rthreads < kScatterThreads) rthreads = kScatterThreads;  // (the cache rows' scatter needs its own threads)
    if (pack > 1) pack = rthreads / kWave;
    const int blocks = (pack > 1 ? (nslices + pack - 1) / pack : nslices) + nsegs;  // slice owners, then the hot slices' segment work-groups
    ProfScope ps(TTX_PROF_APPLY, st);
    CacheTail CT{};
    int tail_blocks = 0;
    if (tail && tail->dst) {  // the cache rows' scatter in the same launch (launch_scatter_add's grid, ttx_cache.hip)
      CT = *tail;
      CT.first = blocks;
      CT.nmain = (int)((CT.N + rthreads / 32 - 1) / (rthreads / 32));  // (a lookup per 32 lanes, all of the work-group's threads)
      CT.K = (((uintptr_t)CT.grad & 15) == 0) ? hot_rows(CT.N, CT.D) : 0;
      tail_blocks = CT.nmain + CT.K * (int)((CT.N + kHotSeg - 1) / kHotSeg);
      if (tail_done) *tail_done = 1;
    }
    hipLaunchKernelGGL(reduce_apply_kernel, dim3(blocks + tail_blocks), dim3(rthreads), reduce_lds_bytes(rthreads), st, d, P, PC, optim, lr,
                       eps, C, S, DW, nslices, (g_skip_launch & 4) ? -(int)nnz : (int)nnz, CT, pack);
    TTX_HIP(hipGetLastError());
  }
  return TTX_OK;
}

int ttx_psw_backward(int32_t B, int32_t D, int64_t nnz, const float* rows, const int64_t* rowidx,
                     const int64_t* tableidx, const float* d_output, float* d_psw, ttx_stream_t stream) {
  (void)hipGetLastError();
  if (nnz == 0) return TTX_OK;
  if (B <= 0 || D <= 0 || nnz < 0 || nnz >= (1ll << 31)) TTX_FAIL(TTX_EINVAL, "bad B / D / nnz");
  if (!rows || !rowidx || !tableidx || !d_output || !d_psw) TTX_FAIL(TTX_EINVAL, "NULL input");
  hipLaunchKernelGGL(psw_grad_kernel, dim3(((int)nnz + kThreads / 16 - 1) / (kThreads / 16)), dim3(kThreads), 0,
                     (hipStream_t)stream, (int)nnz, B, D, rows, rowidx, tableidx, d_output, d_psw);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

// ---- duplicate lookups share their contraction (include/ttx.h) --------------------------------------------
size_t ttx_dedup_bytes(const ttx_geom* g, int64_t nnz) {
  Dims d;
  if (make_dims(g, &d) != TTX_OK || !dedup_supported(d, nnz)) return 0;
  return dedup_bytes(nnz);
}

int ttx_dedup_build(const ttx_geom* g, int64_t nnz, const int64_t* indices, const int64_t* tableidx, void* dedup,
                    size_t dedup_bytes_, void* plan, size_t plan_bytes_, ttx_stream_t stream) {
  Dims d;
  int rc = make_dims(g, &d);
  if (rc) return rc;
  if (!dedup_supported(d, nnz))
    TTX_FAIL(TTX_EUNSUPPORTED, "duplicate sharing: tables of one row shape, tables * prod(p) < 2^61, nnz < 2^31");
  if (!indices || (d.num_tables > 1 && !tableidx)) TTX_FAIL(TTX_EINVAL, "NULL input");
  if (!dedup || dedup_bytes_ < dedup_bytes(nnz)) TTX_FAIL(TTX_EWORKSPACE, "dedup buffer too small: %zu < %zu", dedup_bytes_, dedup_bytes(nnz));
  if (!plan || plan_bytes_ < plan_bytes(d, nnz)) TTX_FAIL(TTX_EWORKSPACE, "plan buffer too small: %zu < %zu", plan_bytes_, plan_bytes(d, nnz));
  rc = common_checks(d, d.D, nnz);
  if (rc) return rc;
  const DedupMap M = carve_dedup(nnz, dedup);
  rc = dedup_build(d, nnz, indices, tableidx, M, (hipStream_t)stream);
  if (rc) return rc;
  // the plan of the DISTINCT pairs: their count lives on the device, "bag row" of pair u is u
  Plan P = carve_plan(d, nnz, plan);
  // (the pairs are table-major: with several tables the plan may sort each table group by itself, as it does for the
  //  module's table-major bags -- "offsets" = first pair of every table, one "bag" per table)
  const bool grouped = d.num_tables <= dedup_max_tables() && plan_groups_tables(d, nnz);
  return plan_build(d, nnz, M.uidx, M.utab, M.iota, P, (hipStream_t)stream, M.nu, grouped ? dedup_tstart(M) : nullptr, 1);
}

size_t ttx_tt_forward_dd_workspace_bytes(const ttx_geom* g, int32_t D, int64_t nnz) {
  (void)D;
  Dims d;
  if (make_dims(g, &d) != TTX_OK || nnz < 0) return 0;
  return rows_bytes(d, nnz) + 256;
}

int ttx_tt_forward_dd(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* rowidx,
                      const int64_t* tableidx, const float* psw, const void* dedup, const void* plan,
                      const float* const* tt_cores, float* output, void* workspace, size_t workspace_bytes,
                      ttx_stream_t stream) {
  Dims d;
  int rc = make_dims(g, &d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (B < 0 || !output) TTX_FAIL(TTX_EINVAL, "bad B/output");
  if (!dedup_supported(d, nnz)) TTX_FAIL(TTX_EUNSUPPORTED, "not a deduplicated batch");
  rc = common_checks(d, D, nnz);
  if (rc) return rc;
  if (!rowidx || !tableidx || !tt_cores || !dedup || !plan) TTX_FAIL(TTX_EINVAL, "NULL input");
  if (!workspace || workspace_bytes < rows_bytes(d, nnz))
    TTX_FAIL(TTX_EWORKSPACE, "forward workspace too small: %zu < %zu", workspace_bytes, rows_bytes(d, nnz));
  const DedupMap M = carve_dedup(nnz, (void*)dedup);
  Plan P = carve_plan(d, nnz, (void*)plan);
  float* rows = (float*)workspace;
  const long long nout = (long long)d.num_tables * B * d.D;
  rc = run_rows(d, nnz, P, tt_cores, rows, output, nout, st);  // rows of the distinct pairs; also zeroes `output`
  if (rc) return rc;
  ProfScope ps(TTX_PROF_POOL, st);
  const int blocks = ((int)nnz + kThreads / 16 - 1) / (kThreads / 16);
  if (d.D % 4 == 0 && (((uintptr_t)rows | (uintptr_t)output) & 15) == 0 && nnz > kPoolSpanMin)
    hipLaunchKernelGGL(pool_gather4_kernel, dim3(((int)nnz + kThreads - 1) / kThreads), dim3(kThreads), 0, st, (int)nnz, B, d.D / 4,
                       rowidx, tableidx, M.uid, (const float4*)rows, psw, (float4*)output);
  else if (d.D % 4 == 0 && (((uintptr_t)rows | (uintptr_t)output) & 15) == 0)
    hipLaunchKernelGGL(pool_gather_kernel<float4>, dim3(blocks), dim3(kThreads), 0, st, (int)nnz, B, d.D / 4, rowidx, tableidx,
                       M.uid, (const float4*)rows, psw, (float4*)output);
  else
    hipLaunchKernelGGL(pool_gather_kernel<float>, dim3(blocks), dim3(kThreads), 0, st, (int)nnz, B, d.D, rowidx, tableidx, M.uid,
                       (const float*)rows, psw, output);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

// summed bag gradients of the distinct pairs [nnz][D] + the slice partials of the pre-sum [slices][2][D]
static size_t gu_rows_bytes(const Dims& d, long long nnz) { return align_up((size_t)nnz * d.D * sizeof(float)); }
static size_t gu_bytes(const Dims& d, long long nnz) {
  const size_t slices = ((size_t)nnz + kGsSlice - 1) / kGsSlice;
  return gu_rows_bytes(d, nnz) + align_up(slices * 2 * d.D * sizeof(float));
}

}  // extern "C"
namespace ttx {
size_t gsum_scratch_bytes(int D, long long nnz) {
  const size_t slices = ((size_t)nnz + kGsSlice - 1) / kGsSlice;
  return align_up(slices * 2 * (size_t)D * sizeof(float));
}
int gsum_launch(const DedupMap& M, long long nnz, int B, int D, const int64_t* rowidx, const int64_t* tableidx, const float* psw,
                const float* d_output, float* Gu, void* scratch, hipStream_t st) {
  const int N = (int)nnz;
  const int blocks = ((N + kGsSlice - 1) / kGsSlice + kGsThreads / 16 - 1) / (kGsThreads / 16);
  float* Pp = (float*)scratch;
  if (D % 4 == 0 && ((((uintptr_t)d_output) | ((uintptr_t)Gu)) & 15) == 0) {
    hipLaunchKernelGGL(gsum_slice_kernel<float4>, dim3(blocks), dim3(kGsThreads), 0, st, M, N, B, D / 4, rowidx, tableidx, psw,
                       (const float4*)d_output, (float4*)Gu, (float4*)Pp);
    hipLaunchKernelGGL(gsum_fold_kernel<float4>, dim3(blocks), dim3(kGsThreads), 0, st, M, N, D / 4, (const float4*)Pp, (float4*)Gu);
  } else {
    hipLaunchKernelGGL(gsum_slice_kernel<float>, dim3(blocks), dim3(kGsThreads), 0, st, M, N, B, D, rowidx, tableidx, psw, d_output,
                       Gu, Pp);
    hipLaunchKernelGGL(gsum_fold_kernel<float>, dim3(blocks), dim3(kGsThreads), 0, st, M, N, D, (const float*)Pp, Gu);
  }
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}
}  // namespace ttx
extern "C" {

size_t ttx_tt_backward_dd_workspace_bytes(const ttx_geom* g, int32_t D, int64_t nnz) {
  Dims d;
  if (make_dims(g, &d) != TTX_OK || nnz < 0) return 0;
  const size_t inner = ttx_tt_backward_workspace_bytes(g, 0, D, nnz);
  return inner ? gu_bytes(d, nnz) + inner : 0;
}

int ttx_tt_backward_dd(const ttx_geom* g, int32_t optim, int32_t B, int32_t D, float lr, float eps, int64_t nnz,
                       const int64_t* rowidx, const int64_t* tableidx, const float* psw, const float* d_output,
                       const void* dedup, const void* plan, float* const* tt_cores, float* const* optimizer_state,
                       float* const* d_tt_cores, void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
  Dims d;
  int rc = make_dims(g, &d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (!dedup_supported(d, nnz)) TTX_FAIL(TTX_EUNSUPPORTED, "not a deduplicated batch");
  rc = common_checks(d, D, nnz);
  if (rc) return rc;
  if (!rowidx || !tableidx || !d_output || !dedup || !plan) TTX_FAIL(TTX_EINVAL, "NULL input");
  const size_t gb = gu_bytes(d, nnz);
  if (!workspace || workspace_bytes < gb) TTX_FAIL(TTX_EWORKSPACE, "backward workspace too small");
  const DedupMap M = carve_dedup(nnz, (void*)dedup);
  float* Gu = (float*)workspace;
  {
    ProfScope ps(TTX_PROF_POOL, st);
    rc = gsum_launch(M, nnz, B, d.D, rowidx, tableidx, psw, d_output, Gu, (char*)workspace + gu_rows_bytes(d, nnz), st);
    if (rc) return rc;
  }
  // the distinct pairs as a batch of their own: bag row of pair u is u, its bag gradient Gu[u] (B = 0: no table term)
  return ttx_tt_backward_w(g, optim, 0, D, lr, eps, nnz, M.uidx, M.iota, M.utab, nullptr, Gu, tt_cores, optimizer_state,
                           d_tt_cores, plan, (char*)workspace + gb, workspace_bytes - gb, stream);
}

}  // extern "C"

