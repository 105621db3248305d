// ttx_internal.h -- shared host/device definitions of libttx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ttx.h"
#ifdef TTX_TEST_HOOKS  // the test build (libttx_hooks.so): knobs are globals with setters; the product build folds them to constants
#include "../../include/ttx_test_hooks.h"
#define TTX_KNOB(type, name, dflt) type name = dflt
#else
#define TTX_KNOB(type, name, dflt) constexpr type name = dflt
#endif

namespace ttx {

constexpr int kWave = 64;  // CDNA wavefront

// ---------------------------------------------------------------- errors ----
void set_error(const char* fmt, ...);
#define TTX_FAIL(code, ...)     \
  do {                          \
    ttx::set_error(__VA_ARGS__); \
    return (code);              \
  } while (0)
#define TTX_HIP(call)                                                         \
  do {                                                                        \
    hipError_t e__ = (call);                                                  \
    if (e__ != hipSuccess)                                                    \
      TTX_FAIL(TTX_EHIP, "%s failed: %s", #call, hipGetErrorString(e__));     \
  } while (0)

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Allow `kernel` to be launched with up to `bytes` of dynamic LDS on the CURRENT device (beyond 64 KiB a kernel needs
// hipFuncAttributeMaxDynamicSharedMemorySize).  The attribute is per device: remembered per (kernel, device) under a
// lock -- the forward and the autograd thread, and several devices of one process, all get here.  TTX_OK / TTX_EHIP.
int allow_dynamic_lds(const void* kernel, int bytes);
int device_cus();  // compute units of the current device

// ------------------------------------------------------------- geometry ----
// Derived per-stage GEMM shapes of the TT chain:
//   x_t[m_t x n_t] = x_{t-1}[m_t x k_t] * core_{t+1}[i_{t+1}][k_t x n_t]
// unsigned 32-bit division by an invariant divisor (Granlund-Montgomery):
//   q = (t + ((n - t) >> sh1)) >> sh2,  t = mulhi(m, n)   exact for all n < 2^32
struct UDiv {
  unsigned m, sh1, sh2, d;
};

// tables of different row factors in one batch (ttx_geom::p_tables): device-resident, one per distinct
// geometry (cached by make_dims, never freed: a few KiB each)
struct TabGeom {
  int p[TTX_MAX_TABLES_MIXED][TTX_MAX_CORES];
  int base[TTX_MAX_TABLES_MIXED][TTX_MAX_CORES];  // first slice of the table in core t
  long long L[TTX_MAX_TABLES_MIXED][TTX_MAX_CORES];
};

struct Dims {
  int T, num_tables;
  const TabGeom* tab;        // NULL unless the tables differ in p (then p[] = the largest, S[] = the sums)
  int p[TTX_MAX_CORES], q[TTX_MAX_CORES], r[TTX_MAX_CORES + 1];
  long long L[TTX_MAX_CORES];
  int slice[TTX_MAX_CORES];  // r_t q_t r_{t+1}
  int S[TTX_MAX_CORES];      // num_tables * p_t  (slices of core t)
  int m[TTX_MAX_CORES], k[TTX_MAX_CORES], n[TTX_MAX_CORES];
  int D;
  int idx32;                 // prod(p) <= 2^32: indices decode with 32-bit magic division
  UDiv dvL[TTX_MAX_CORES], dvP[TTX_MAX_CORES];
};

int make_dims(const ttx_geom* g, Dims* d);  // TTX_OK or TTX_EINVAL (+message)

// ----------------------------------------------------------------- plan ----
// Device-resident lookup plan (see include/ttx.h).  All arrays int32.
//   sid[t][n]   = table*p_t + i_t                      (original order)
//   perm[t][*]  = lookups n sorted (stably) by sid[t]
//   ipos[t][n]  = position of lookup n in perm[t]: the row of its partial gradient (the backward kernels store
//                 thin-core partials in SORTED order, so a slice's rows are contiguous for reduce_apply)
//   off[t][s]   = first position in perm[t] of slice s  (S_t + 1 entries)
//   chunk_rec[c]= {pivot slice, start in the sorted order, lookups in chunk, partial slot}: the work
//                 list of the pivot core (core 1), each slice's run of lookups
//                 cut into chunks of <= MC lookups; chunk_off[s] = first chunk of s
//   lrec[i]     = {n, sid_0, sid_2, sid_3} of the i-th lookup in pivot order
//   lrow[i]     = bag row (rowidx[n]) of that lookup; valid iff hdr[3] != 0
// reduce_apply_kernel's hot-slice thresholds (ttx_tt.hip); the plan kernels that know the slice sizes leave the number
// of hot slices per core in hdr[8 + t] (-1: unknown), so that the launch's hot-slice work-groups can leave at once
#ifndef TTX_SEG_THIN
#define TTX_SEG_THIN 256
#endif
#ifndef TTX_HOT_PIVOT
#define TTX_HOT_PIVOT 16
#endif
constexpr int kSegThin = TTX_SEG_THIN;    // partial rows per segment of a thin core's sorted order; hot: > 2 segments
constexpr int kHotRowsPivot = TTX_HOT_PIVOT;  // chunk partials beyond which a pivot slice is hot

// hdr[kHdrT4Valid .. +4]: four cores on the three-core kernels -- 1 while Plan::t4m / t4o hold the merged last cores M of THIS plan's
// lookups for the core tensors whose addresses follow (two ints each: core 2, core 3).  Set by the forward's merge, read by the
// backward's (which then skips its own), cleared by every plan build and by the fused optimizer's write to cores 2 / 3; [+5] = the
// device-wide count of such writes (g_t4_epoch, ttx_tt.hip) when M was made: another plan's fused backward invalidates this M too.
constexpr int kHdrT4Valid = 20;
// hdr[kHdrGrab]: the chunk counter of bwd32_kernel's persistent work-groups (ttx_tt_spec.inc), zeroed in front of its launch
constexpr int kHdrGrab = 32;
struct Plan {
  int* hdr;  // [0] = number of chunks, [1] = MC, [2] = nnz, [3] = lrow valid, [8 + t] = hot slices of core t (-1: unknown), [20..25]: kHdrT4Valid
  int* sid[TTX_MAX_CORES];
  int* perm[TTX_MAX_CORES];
  int* ipos[TTX_MAX_CORES];  // inverse of perm: position of lookup n in core t's sorted order (thin cores)
  int* off[TTX_MAX_CORES];
  int* chunk_off;    // [S_1 + 1]
  int4* chunk_rec;   // [max_chunks]
  int4* lrec;        // [nnz]
  int* lrow;         // [nnz]
  int* cnt;          // multi-block sort: [T][256][units]
  int* scratch[TTX_MAX_CORES][3];  // rank / ping / pong per core, [nnz] each
  int max_chunks;
  int MC;
  // four cores through the three-core kernels (t4_scratch_floats > 0): per lookup, in pivot order, the last two cores'
  // product M [r2][q2 q3] and, backward, its gradient
  float* t4m;   // [position][r2 q2 q3]
  float* t4g;   // [lookup n][r2 q2 q3]
  int4* t4o;    // [position in core 2's sorted order] = {lookup n, sid_2, sid_3, row of its core-3 partial (ipos[3][n])}
};

int choose_chunk(const Dims& d, long long nnz);  // lookups per chunk (kernel variant / LDS-budget heuristic)
// floats of per-lookup scratch a four-core geometry needs in its plan to run on the three-core kernels (0: it does not)
size_t t4_scratch_floats(const Dims& d);
int max_chunks(const Dims& d, long long nnz, int MC);
size_t plan_bytes(const Dims& d, long long nnz);
// carve `base` into the plan arrays (same function for builder and consumers)
Plan carve_plan(const Dims& d, long long nnz, void* base);
int plan_build(const Dims& d, long long nnz, const int64_t* indices,
               const int64_t* tableidx, const int64_t* rowidx, const Plan& P, hipStream_t stream,
               const int* n_dev = nullptr, const int64_t* offsets = nullptr, int bags_per_table = 0);
// (offsets != NULL: the caller vouches that the lookups are table-major with table k at positions
//  [offsets[k * bags_per_table], offsets[(k + 1) * bags_per_table)) -- the module's bag offsets)  // n_dev: device-side lookup count <= nnz (nnz is then an upper bound)

// Several batches in ONE launch (ttx_lookup_prologue_multi / _cached_multi: a round of training batches planned ahead):
// grid.z (or .y) = batch.  Batch z reads its own indices / offsets and writes its outputs at a constant stride behind
// batch 0's.
constexpr int kMaxMulti = 16;
struct ProBatch {
  const int64_t* indices[kMaxMulti];
  const int64_t* offsets[kMaxMulti];
  long long out_stride;    // elements between the batches' rowidx / tableidx (/ pcol / prow / ploc) arrays
  long long plan_stride;   // bytes between the batches' plan buffers
};
// indices / offsets of batch Z > 0.  (A switch over constant subscripts: a run-time subscript into the by-value struct
// sends it -- and the kernel's other arguments with it -- through scratch memory: measured +10 us on every launch.)
#define TTX_PICK_BATCH(MB, Z, IND, OFF)                                                                   \
  switch (Z) {                                                                                            \
    case 1: IND = MB.indices[1]; OFF = MB.offsets[1]; break;                                              \
    case 2: IND = MB.indices[2]; OFF = MB.offsets[2]; break;                                              \
    case 3: IND = MB.indices[3]; OFF = MB.offsets[3]; break;                                              \
    case 4: IND = MB.indices[4]; OFF = MB.offsets[4]; break;                                              \
    case 5: IND = MB.indices[5]; OFF = MB.offsets[5]; break;                                              \
    case 6: IND = MB.indices[6]; OFF = MB.offsets[6]; break;                                              \
    case 7: IND = MB.indices[7]; OFF = MB.offsets[7]; break;                                              \
    case 8: IND = MB.indices[8]; OFF = MB.offsets[8]; break;                                              \
    case 9: IND = MB.indices[9]; OFF = MB.offsets[9]; break;                                              \
    case 10: IND = MB.indices[10]; OFF = MB.offsets[10]; break;                                           \
    case 11: IND = MB.indices[11]; OFF = MB.offsets[11]; break;                                           \
    case 12: IND = MB.indices[12]; OFF = MB.offsets[12]; break;                                           \
    case 13: IND = MB.indices[13]; OFF = MB.offsets[13]; break;                                           \
    case 14: IND = MB.indices[14]; OFF = MB.offsets[14]; break;                                           \
    case 15: IND = MB.indices[15]; OFF = MB.offsets[15]; break;                                           \
    default: break;                                                                                       \
  }
bool plan_batches_ok(const Dims& d, long long nnz);
int plan_build_batches(const Dims& d, int nbatch, long long nnz, const int* n_dev, const int64_t* indices,
                       const int64_t* tableidx, const int64_t* rowidx, void* plans, size_t plan_stride,
                       hipStream_t stream);

long long* debug_stamps();  // debug stamp buffer (ttx_debug_stamps), or nullptr

// --------------------------------------------------- duplicate lookups ----
// Device-resident map of a batch onto its DISTINCT (table, index) pairs (include/ttx.h, ttx_dedup_build): the
// contraction kernels then run once per distinct pair, bag pooling gathers a lookup's row through uid[], and the
// backward first sums the bag gradients of a pair's occurrences (in index order) into one row.
//   nu         = number of distinct pairs                                   (device int)
//   uidx/utab  = the distinct pairs, ascending (table, index)               [nnz], first nu valid
//   iota       = 0..nnz-1: "bag row" u of distinct pair u                   [nnz]
//   uid[n]     = distinct pair of lookup n                                  [nnz]
//   occ        = lookups sorted by (uid, n); occ_off[u] = first position of pair u, occ_off[nu] = nnz
struct DedupMap {
  int* nu;
  int64_t* uidx;
  int64_t* utab;
  int64_t* iota;
  int* uid;
  int* occ;
  int* occ_off;
};
constexpr int kDedupMaxN = 16384;  // one work-group sorts the batch in LDS
// header of the map: int [0] = distinct pairs; from byte 64 on, int64 tstart[tables + 1] = first pair of every table (the pairs
// are table-major; written for 1 < tables <= dedup_max_tables(), the "offsets" the plan of the pairs groups its tables by)
__host__ __device__ inline int64_t* dedup_tstart(const DedupMap& M) { return (int64_t*)((char*)M.nu + 64); }
int dedup_max_tables();
bool plan_groups_tables(const Dims& d, long long N);  // would plan_build use table groups, given table-major offsets?
bool dedup_supported(const Dims& d, long long nnz);
size_t dedup_bytes(long long nnz);
DedupMap carve_dedup(long long nnz, void* base);
int dedup_build(const Dims& d, long long nnz, const int64_t* indices, const int64_t* tableidx, const DedupMap& M,
                hipStream_t stream);
// the map over the caller's own 64-bit keys (ttx_plan.hip; used by the sorted cache-row update, ttx_cache.hip)
void dedup_key_buffers(const DedupMap& M, long long nnz, int64_t** keys, int64_t** vals);
int dedup_build_from_keys(long long nnz, unsigned long long all, const DedupMap& M, hipStream_t stream);
// Gu[u, :] = sum over the occurrences n of distinct key u of (psw[n] *) d_output[(tableidx[n] * B +) rowidx[n], :], in index order,
// no atomics (ttx_tt.hip gsum_slice_kernel / gsum_fold_kernel).  scratch: gsum_scratch_bytes(D, nnz) bytes.
size_t gsum_scratch_bytes(int D, long long nnz);
int gsum_launch(const DedupMap& M, long long nnz, int B, int D, const int64_t* rowidx, const int64_t* tableidx, const float* psw,
                const float* d_output, float* Gu, void* scratch, hipStream_t st);

// stable DESCENDING LSD radix sort of (int64 key, int64 value) pairs, 8 bits per pass, multi-work-group (ttx_cache.hip):
// cache_populate's sort of (frequency, key) and the key sort of the duplicate map.  ws: sort_pairs_ws_bytes(n);
// passes: number of 8-bit passes, or -1 to size the sort by one read-back of the largest key (not capturable).
size_t sort_pairs_ws_bytes(int64_t n);
int sort_pairs_desc(int64_t n, const int64_t* keys_in, const int64_t* vals_in, char* ws, int64_t** keys_sorted,
                    int64_t** vals_sorted, hipStream_t st, int passes);

// ------------------------------------------------------------ profiling ----
void prof_begin(int which, hipStream_t s);
void prof_end(int which, hipStream_t s);
struct ProfScope {
  int w;
  hipStream_t s;
  ProfScope(int which, hipStream_t st) : w(which), s(st) { prof_begin(w, s); }
  ~ProfScope() { prof_end(w, s); }
};

// ------------------------------------------------------- device helpers ----
#ifdef __HIPCC__
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ unsigned udiv(unsigned n, const UDiv dv) {
  const unsigned t = __umulhi(dv.m, n);
  return (t + ((n - t) >> dv.sh1)) >> dv.sh2;
}

// peers of this lane: valid lanes of the wave holding the same 8-bit digit.
// Built from 9 wave ballots (the gfx950 replacement for a CUB rank pass).
// Must be called by all lanes of the wave (wave-uniform control flow).
template <int BITS>
__device__ __forceinline__ unsigned long long wave_match(unsigned dgt, bool valid) {
  unsigned long long mask = __ballot(valid);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    const bool bit = (dgt >> b) & 1u;
    const unsigned long long bm = __ballot(bit && valid);
    mask &= bit ? bm : ~bm;
  }
  return mask;
}
__device__ __forceinline__ unsigned long long wave_match8(unsigned dgt, bool valid) { return wave_match<8>(dgt, valid); }

__device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}

// wave-wide inclusive scan (sum) via DPP-friendly shuffles
__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    int u = __shfl_up(v, o, kWave);
    if (lane_id() >= o) v += u;
  }
  return v;
}
constexpr int kMaxProbes = 3;  // tt_embeddings_cuda.cu:29

// hashtbl_cuda_utils.cuh:48-76 (bit-exact restatement; KATs in tests/golden)
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ __forceinline__ uint32_t hash64(int64_t key, int32_t C) {
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  const uint64_t u = (uint64_t)key;
  uint32_t h = 0;
  uint32_t k1 = (uint32_t)u;
  k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
  h ^= k1; h = rotl32(h, 13); h = h * 5 + 0xe6546b64u;
  uint32_t k2 = (uint32_t)(u >> 32);
  k2 *= c1; k2 = rotl32(k2, 15); k2 *= c2;
  h ^= k2; h = rotl32(h, 13); h = h * 5 + 0xe6546b64u;
  h ^= 2;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return (uint32_t)(((uint64_t)h * (uint64_t)(uint32_t)C) >> 32);
}

// hashtbl_cuda_utils.cuh:102-133 with accumulate == true: a 64-bit CAS claims the slot (or finds
// the key already there), a 64-bit atomic add bumps its frequency; dropped after kMaxProbes.
// -> the slot the key was counted in, -1 if it was dropped.  That slot is the first one of the key's probe sequence that holds
// the key once this launch's inserts are done -- i.e. what hashtbl_find (ttx_cache.hip) returns AFTER the update, for every copy
// of the key in the launch alike (a slot seen holding another key keeps it; an empty one goes to whoever's CAS lands, and every
// later CAS on it returns that winner).
__device__ __forceinline__ int32_t hashtbl_count(int64_t key, int32_t H, int64_t* hashtbl, int64_t* cache_freq,
                                                 unsigned long long times = 1ull) {
  int32_t idx = (int32_t)hash64(key, H);
  for (int c = 0; c < kMaxProbes; ++c) {
    // A slot only ever goes from empty to a key while inserts run (eviction is cache_populate's, another launch), so a
    // plain load that shows a key is final: the CAS is issued only where the slot looks empty -- in steady state, where
    // nearly every key is already seated, that leaves ONE atomic per key (the count) instead of two.  (A stale "empty"
    // just takes the CAS.)
    unsigned long long old = (unsigned long long)hashtbl[idx];
    if ((int64_t)old == -1)
      old = atomicCAS((unsigned long long*)&hashtbl[idx], (unsigned long long)(-1ll), (unsigned long long)key);
    if ((int64_t)old == -1 || (int64_t)old == key) {
      atomicAdd((unsigned long long*)&cache_freq[idx], times);
      return idx;
    }
    idx = (idx + 1) % H;
  }
  return -1;
}

// The same for one key per lane, equal keys of a wave combined first: under a skewed index stream a
// third of a batch is ONE key, and thousands of CAS + add on one slot serialise in L2 (measured: the
// 10k-key update took 33 us instead of 5).  Lanes are grouped by the low 8 bits of the key's hash
// (9 ballots); the lowest lane of a group leads; lanes holding the leader's key hand it their count,
// the rare others (8-bit collisions) go alone.  Same table contents as one insert per lane (counts
// add up; a dropped key is dropped for all its copies either way).  Wave-uniform call.  -> the lane's slot (hashtbl_count above;
// the lanes that handed their count to a leader get the leader's), -1 for a dropped key or an invalid lane.
__device__ __forceinline__ int32_t hashtbl_count_wave(int64_t key, bool valid, int32_t H, int64_t* hashtbl,
                                                      int64_t* cache_freq) {
  const unsigned h = valid ? hash64(key, H) : 0u;
  const unsigned long long peers = wave_match8(h & 255u, valid);
  const int leader = valid ? __ffsll((long long)peers) - 1 : 0;
  const int klo = __shfl((int)(unsigned)key, leader, kWave), khi = __shfl((int)(key >> 32), leader, kWave);
  const bool eq = valid && klo == (int)(unsigned)key && khi == (int)(key >> 32);
  const unsigned long long eqm = __ballot(eq);
  int32_t slot = -1;
  if (valid) {
    if (!eq) slot = hashtbl_count(key, H, hashtbl, cache_freq);
    else if (lane_id() == leader) slot = hashtbl_count(key, H, hashtbl, cache_freq, (unsigned long long)__popcll(peers & eqm));
  }
  const int32_t ls = __shfl(slot, leader, kWave);  // (all lanes: the leaders' slots for the lanes they counted for)
  return eq ? ls : slot;
}
#endif

}  // namespace ttx
