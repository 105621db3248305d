// ttx_tt_common.h -- what the contraction kernels' translation units share: ttx_tt.hip (generic kernels, pooling,
// reduce + apply, the host side) and ttx_tt_spec{16,32,64,128a,128b,128c}.hip (the shape-specialised kernels, one rank family per
// translation unit so that they compile in parallel -- __graft_entry__.build()).  Inside namespace ttx.
#pragma once
#include "ttx_internal.h"

namespace ttx {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / kWave;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct CorePtrs {
  float* c[TTX_MAX_CORES];
};

// n / d and n % d for small operands via one fp32 multiply and a fix-up
struct FastDiv {
  unsigned d;
  float rcp;
};
__host__ __device__ __forceinline__ FastDiv make_fd(int d) {
  FastDiv f;
  f.d = (unsigned)(d > 0 ? d : 1);
  f.rcp = 1.0f / (float)f.d;
  return f;
}
#define make_fd_dev make_fd
__device__ __forceinline__ unsigned fdivmod(unsigned n, const FastDiv f, unsigned& rem) {
  if (n >> 22) {
    rem = n % f.d;
    return n / f.d;
  }
  unsigned q = (unsigned)((float)n * f.rcp);
  int r = (int)n - (int)(q * f.d);
  if (r < 0) { q -= 1; r += (int)f.d; }
  else if (r >= (int)f.d) { q += 1; r -= (int)f.d; }
  rem = (unsigned)r;
  return q;
}

struct Partials {
  float* pc[TTX_MAX_CORES];  // pc[1] is per CHUNK, the others per lookup
  const float* psw;          // per_sample_weights by lookup (nn.EmbeddingBag), or NULL: the bag gradient of
                             // lookup n enters the backward scaled by psw[n]
  const int64_t* tableidx;   // tables of different row factors (Dims::tab): the table of a pivot slice is read
                             // from one of its lookups; NULL otherwise (table = slice / p_1)
  // hot slices (reduce_apply_kernel): arrival counters, one per core slice (zeroed by the backward
  // contraction kernel), and the segment partial sums, two slots per segment
  int* hot_cnt;
  int n_hot_cnt;
  float* seg[TTX_MAX_CORES];
};

// zero the hot-slice arrival counters (called by work-group 0 of the backward contraction kernels,
// which always run right before reduce_apply_kernel on the same stream)
__device__ __forceinline__ void zero_hot_counters(const Partials& PC) {
  if (blockIdx.x == 0 && PC.hot_cnt)
    for (int i = threadIdx.x; i < PC.n_hot_cnt; i += blockDim.x) PC.hot_cnt[i] = 0;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// sixteen independent 4 x 4 outer products (a block = four consecutive lanes): D[lane 4b+j][reg i] += a[lane 4b+i] * b[lane 4b+j]
__device__ __forceinline__ f32x4 mfma1(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// the grid also zeroes the pooled output (the bag-pooling kernel that follows accumulates
// onto it): saves a memset launch
__device__ __forceinline__ void zero_output(float* __restrict__ out, long long n) {
  if (!out) return;
  for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < n; e += (long long)gridDim.x * kThreads) out[e] = 0.f;
}

// test / ablation knobs (include/ttx_test_hooks.h).  They exist in the TEST build only (-DTTX_TEST_HOOKS -> libttx_hooks.so): there
// they are globals of the library with setters (ttx_debug_skip, ttx_debug_stamps, ...), defined in ttx_tt.hip.  In the product
// build (libttx.so) every knob is a compile-time constant at its default and no setter is compiled.
#ifdef TTX_TEST_HOOKS
extern int g_debug_skip;
extern int g_disable_spec;
extern long long* g_stamps;
extern int g_bwd32_mc;  // ttx_debug_bwd32: lookups per chunk of bwd32_kernel (ttx_tt_spec.inc), 0 = spec_bwd_kernel as in the product
#else
constexpr int g_debug_skip = 0;
constexpr int g_disable_spec = 0;
constexpr long long* g_stamps = nullptr;
constexpr int g_bwd32_mc = 0;
#endif

}  // namespace ttx
