// ttx_tt.hip -- TT-core contraction forward / backward for gfx950.
//
// Design: lookups are grouped by the slice of the PIVOT core
// (core 1, the big r1 x q1 x r2 slice) they touch.  One work-group owns a chunk
// of <= MC lookups of one pivot slice:
//   forward   X0[MC*q0 x N1] = A[MC*q0 x r1] * B1[r1 x N1]   (B1 staged ONCE in LDS,
//             A = the chunk's core-0 slices stacked), then the remaining
//             T-2 stages per lookup out of LDS, rows -> HBM, bags pooled by a
//             second tiny kernel in index order (same order as the reference's
//             reduce_output_kernel, tt_embeddings_cuda.cu:920-962);
//   backward  recompute X0, per-lookup tail (grad of the last cores), then two
//             chunk GEMMs  dB1 = A^T * dX0   and   dA = dX0 * B1^T.
// Gradients never use atomics: every producer writes a private partial
// (per lookup for the thin cores, per chunk for the pivot) and ONE owner per
// core slice sums them in index order and applies DENSE / SGD / Adagrad
// (reduce_apply_kernel) -- deterministic, and the optimizer touches only
// slices that were looked up.
#include "ttx_internal.h"

namespace ttx {

constexpr int kThreads = 256;

struct CorePtrs {
  float* c[TTX_MAX_CORES];
};

// LDS carve of the contraction kernels (float offsets)
struct Lds {
  int MC;
  int ldA;   // row stride of A  (>= r1)
  int ldB;   // row stride of B1 / X0 rows (>= N1)
  int oB, oA, oX0, oX1, oG, oI;
  int szX0, szX1;  // per-lookup floats of X0 / X1
  int bytes;
};

static Lds make_lds(const Dims& d, int MC, bool bwd) {
  Lds L;
  memset(&L, 0, sizeof(L));
  L.MC = MC;
  const int K0 = d.k[0], N1 = d.n[0], q0 = d.q[0];
  L.ldA = K0;
  L.ldB = N1;
  L.szX0 = q0 * L.ldB;
  L.szX1 = (d.T == 4) ? d.m[1] * d.n[1] : 0;
  int o = 0;
  auto take = [&](int n) { int r = o; o += (n + 3) / 4 * 4; return r; };
  L.oB = take(K0 * L.ldB);
  L.oA = take(MC * q0 * L.ldA);
  L.oX0 = take(MC * L.szX0);
  L.oX1 = take(MC * L.szX1);
  L.oG = take(bwd ? MC * d.D : 0);
  L.oI = take(MC * (1 + TTX_MAX_CORES));
  L.bytes = o * 4;
  return L;
}

static int g_chunk_override = 0;

int choose_chunk(const Dims& d) {
  if (g_chunk_override > 0) return g_chunk_override;
  // prefer <= 64 KiB (two work-groups per CU), else up to the full 160 KiB
  for (int mc = 32; mc >= 8; mc >>= 1)
    if (make_lds(d, mc, true).bytes <= 64 * 1024) return mc;
  for (int mc = 16; mc >= 1; mc >>= 1)
    if (make_lds(d, mc, true).bytes <= 160 * 1024) return mc;
  return 0;
}

// ------------------------------------------------------------- kernels -----

// chunk prologue shared by forward and backward: stage B1, the lookup ids and
// the stacked core-0 slices in LDS.
__device__ __forceinline__ void stage_chunk(const Dims& d, const Plan& P, const CorePtrs& C,
                                            const Lds& L, float* smem, int s, int start, int len) {
  const int tid = threadIdx.x;
  const int K0 = d.k[0], N1 = d.n[0], q0 = d.q[0];
  int* I = (int*)(smem + L.oI);
  if (tid < len) {
    const int n = P.perm[1][start + tid];
    I[tid] = n;
#pragma unroll
    for (int t = 0; t < TTX_MAX_CORES; ++t)
      if (t < d.T) I[(1 + t) * L.MC + tid] = P.sid[t][n];
  }
  const float* B1 = C.c[1] + (size_t)s * d.slice[1];
  float* Bs = smem + L.oB;
  for (int e = tid; e < K0 * N1; e += kThreads) Bs[(e / N1) * L.ldB + (e % N1)] = B1[e];
  __syncthreads();
  float* As = smem + L.oA;
  const int sl0 = d.slice[0];  // q0 * r1
  for (int e = tid; e < len * sl0; e += kThreads) {
    const int j = e / sl0, rem = e % sl0;
    const float* a = C.c[0] + (size_t)I[L.MC + j] * sl0;
    As[(j * q0 + rem / K0) * L.ldA + (rem % K0)] = a[rem];
  }
  __syncthreads();
}

// X0[rows x N1] = As[rows x K0] * Bs[K0 x N1]
__device__ __forceinline__ void gemm_x0(const Dims& d, const Lds& L, float* smem, int rows) {
  const int K0 = d.k[0], N1 = d.n[0];
  const float* As = smem + L.oA;
  const float* Bs = smem + L.oB;
  float* X0 = smem + L.oX0;
  for (int e = threadIdx.x; e < rows * N1; e += kThreads) {
    const int row = e / N1, col = e % N1;
    float acc = 0.f;
    for (int k = 0; k < K0; ++k) acc = fmaf(As[row * L.ldA + k], Bs[k * L.ldB + col], acc);
    X0[row * L.ldB + col] = acc;
  }
}

__global__ __launch_bounds__(kThreads) void fwd_kernel(Dims d, Plan P, CorePtrs C,
                                                      float* __restrict__ rows, Lds L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int chunk = blockIdx.x;
  if (chunk >= P.hdr[0]) return;
  const int s = P.chunk_slice[chunk];
  const int start = P.chunk_start[chunk];
  const int len = min(L.MC, P.off[1][s + 1] - start);
  const int tid = threadIdx.x;
  const int q0 = d.q[0];
  stage_chunk(d, P, C, L, smem, s, start, len);
  gemm_x0(d, L, smem, len * q0);
  __syncthreads();
  const int* I = (const int*)(smem + L.oI);
  const float* Xin = smem + L.oX0;
  int szin = L.szX0;
  for (int t = 1; t <= d.T - 2; ++t) {
    const int mt = d.m[t], kt = d.k[t], nt = d.n[t];
    const bool last = (t == d.T - 2);
    float* Xout = smem + L.oX1;
    const int per = mt * nt;
    for (int e = tid; e < len * per; e += kThreads) {
      const int j = e / per, rem = e % per;
      const int row = rem / nt, col = rem % nt;
      const float* ct = C.c[t + 1] + (size_t)I[(2 + t) * L.MC + j] * d.slice[t + 1];
      const float* xi = Xin + j * szin + row * kt;
      float acc = 0.f;
      for (int k = 0; k < kt; ++k) acc = fmaf(xi[k], ct[k * nt + col], acc);
      if (last) rows[(size_t)I[j] * d.D + rem] = acc;
      else Xout[j * L.szX1 + rem] = acc;
    }
    __syncthreads();
    Xin = Xout;
    szin = L.szX1;
  }
  if (d.T == 2) {
    const int per = d.D;  // q0 * q1
    const int N1 = d.n[0];
    for (int e = tid; e < len * per; e += kThreads) {
      const int j = e / per, rem = e % per;
      rows[(size_t)I[j] * d.D + rem] = Xin[j * L.szX0 + (rem / N1) * L.ldB + (rem % N1)];
    }
  }
}

// out[table,row,:] += sum of the run's rows, in index order (run = consecutive
// lookups with equal (rowidx, tableidx); reference reduce_output_kernel
// cu:920-962).  One 32-lane group per lookup; only run heads work.
__global__ __launch_bounds__(kThreads) void pool_kernel(int N, int B, int D,
                                                       const int64_t* __restrict__ rowidx,
                                                       const int64_t* __restrict__ tableidx,
                                                       const float* __restrict__ rows,
                                                       float* __restrict__ out) {
  const int n = blockIdx.x * (kThreads / 32) + threadIdx.x / 32;
  const int l = threadIdx.x & 31;
  if (n >= N) return;
  const int64_t r = rowidx[n], tb = tableidx[n];
  if (n > 0 && rowidx[n - 1] == r && tableidx[n - 1] == tb) return;
  int sl = 1;
  while (n + sl < N && rowidx[n + sl] == r && tableidx[n + sl] == tb) ++sl;
  float* o = out + ((size_t)tb * B + r) * D;
  for (int e = l; e < D; e += 32) {
    float acc = o[e];
    for (int j = 0; j < sl; ++j) acc += rows[(size_t)(n + j) * D + e];
    o[e] = acc;
  }
}

struct Partials {
  float* pc[TTX_MAX_CORES];  // pc[1] is per CHUNK, the others per lookup
};

__global__ __launch_bounds__(kThreads) void bwd_kernel(Dims d, Plan P, CorePtrs C, int B,
                                                      const int64_t* __restrict__ rowidx,
                                                      const float* __restrict__ d_output,
                                                      Partials PC, Lds L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int chunk = blockIdx.x;
  if (chunk >= P.hdr[0]) return;
  const int s = P.chunk_slice[chunk];
  const int start = P.chunk_start[chunk];
  const int len = min(L.MC, P.off[1][s + 1] - start);
  const int tid = threadIdx.x;
  const int T = d.T, q0 = d.q[0], K0 = d.k[0], N1 = d.n[0], D = d.D;
  stage_chunk(d, P, C, L, smem, s, start, len);
  const int* I = (const int*)(smem + L.oI);
  float* X0 = smem + L.oX0;
  float* X1 = smem + L.oX1;
  float* Gb = smem + L.oG;
  const int table = s / d.p[1];

  if (T == 2) {
    // dX0 is the bag gradient itself: [q0 x q1]
    for (int e = tid; e < len * D; e += kThreads) {
      const int j = e / D, rem = e % D;
      const float g = d_output[((size_t)table * B + rowidx[I[j]]) * D + rem];
      X0[j * L.szX0 + (rem / N1) * L.ldB + (rem % N1)] = g;
    }
    __syncthreads();
  } else {
    // bag gradients of the chunk's lookups
    for (int e = tid; e < len * D; e += kThreads) {
      const int j = e / D, rem = e % D;
      Gb[j * D + rem] = d_output[((size_t)table * B + rowidx[I[j]]) * D + rem];
    }
    // recompute the forward intermediates x_0 (.. x_{T-3})
    gemm_x0(d, L, smem, len * q0);
    __syncthreads();
    if (T == 4) {
      const int mt = d.m[1], kt = d.k[1], nt = d.n[1], per = mt * nt;
      for (int e = tid; e < len * per; e += kThreads) {
        const int j = e / per, rem = e % per, row = rem / nt, col = rem % nt;
        const float* ct = C.c[2] + (size_t)I[3 * L.MC + j] * d.slice[2];
        const float* xi = X0 + j * L.szX0 + row * kt;
        float acc = 0.f;
        for (int k = 0; k < kt; ++k) acc = fmaf(xi[k], ct[k * nt + col], acc);
        X1[j * L.szX1 + rem] = acc;
      }
      __syncthreads();
    }
    // tail stages t = T-2 .. 1
    for (int t = T - 2; t >= 1; --t) {
      const int mt = d.m[t], kt = d.k[t], nt = d.n[t];
      float* Xin = (t == 1) ? X0 : X1;           // x_{t-1}: [mt x kt] per lookup
      const int szin = (t == 1) ? L.szX0 : L.szX1;
      const float* Gin = (t == T - 2) ? Gb : X1;  // d x_t : [mt x nt] per lookup
      const int szg = (t == T - 2) ? D : L.szX1;
      // (a) d core_{t+1}[i_{t+1}] partial = x_{t-1}^T * G   -> HBM, per lookup
      {
        const int per = kt * nt;
        float* pc = PC.pc[t + 1];
        for (int e = tid; e < len * per; e += kThreads) {
          const int j = e / per, rem = e % per, kk = rem / nt, col = rem % nt;
          const float* xi = Xin + j * szin + kk;
          const float* gi = Gin + j * szg + col;
          float acc = 0.f;
          for (int r = 0; r < mt; ++r) acc = fmaf(xi[r * kt], gi[r * nt], acc);
          pc[(size_t)I[j] * d.slice[t + 1] + rem] = acc;
        }
      }
      __syncthreads();
      // (b) d x_{t-1} = G * core_{t+1}[i_{t+1}]^T, over x_{t-1} in place
      {
        const int per = mt * kt;
        for (int e = tid; e < len * per; e += kThreads) {
          const int j = e / per, rem = e % per, row = rem / kt, kk = rem % kt;
          const float* ct = C.c[t + 1] + (size_t)I[(2 + t) * L.MC + j] * d.slice[t + 1] + kk * nt;
          const float* gi = Gin + j * szg + row * nt;
          float acc = 0.f;
          for (int c = 0; c < nt; ++c) acc = fmaf(gi[c], ct[c], acc);
          Xin[j * szin + rem] = acc;
        }
      }
      __syncthreads();
    }
  }
  // chunk GEMMs on dX0 [rows x N1]
  const int rowsM = len * q0;
  const float* As = smem + L.oA;
  const float* Bs = smem + L.oB;
  {
    // d core_1[slice] partial = As^T * dX0  -> HBM, per chunk
    float* pc = PC.pc[1] + (size_t)chunk * d.slice[1];
    for (int e = tid; e < K0 * N1; e += kThreads) {
      const int kk = e / N1, col = e % N1;
      float acc = 0.f;
      for (int r = 0; r < rowsM; ++r) acc = fmaf(As[r * L.ldA + kk], X0[r * L.ldB + col], acc);
      pc[e] = acc;
    }
  }
  {
    // d core_0[i_0] partial = dX0 * Bs^T  -> HBM, per lookup ([q0 x r1] rows)
    float* pc = PC.pc[0];
    const int sl0 = d.slice[0];
    for (int e = tid; e < len * sl0; e += kThreads) {
      const int j = e / sl0, rem = e % sl0, a = rem / K0, kk = rem % K0;
      const float* xr = X0 + (j * q0 + a) * L.ldB;
      const float* br = Bs + kk * L.ldB;
      float acc = 0.f;
      for (int c = 0; c < N1; ++c) acc = fmaf(xr[c], br[c], acc);
      pc[(size_t)I[j] * sl0 + rem] = acc;
    }
  }
}

// one work-group per core slice: sum the slice's partials in index order and
// apply.  DENSE writes the gradient (zeros for untouched slices: no memset of
// d_tt_cores is needed); SGD / ADAGRAD skip untouched slices (g == 0).
__global__ __launch_bounds__(kThreads) void reduce_apply_kernel(Dims d, Plan P, Partials PC,
                                                               int optim, float lr, float eps,
                                                               CorePtrs W, CorePtrs St,
                                                               CorePtrs DW) {
  int b = blockIdx.x;
  int t = 0;
  while (t < d.T - 1 && b >= d.S[t]) { b -= d.S[t]; ++t; }
  const int s = b;
  const int sl = d.slice[t];
  int beg, end;
  const int* list;
  if (t == 1) { beg = P.chunk_off[s]; end = P.chunk_off[s + 1]; list = nullptr; }
  else { beg = P.off[t][s]; end = P.off[t][s + 1]; list = P.perm[t]; }
  const size_t base = (size_t)s * sl;
  if (beg == end) {
    if (optim == TTX_OPTIM_DENSE)
      for (int e = threadIdx.x; e < sl; e += kThreads) DW.c[t][base + e] = 0.f;
    return;
  }
  const float* pc = PC.pc[t];
  for (int e = threadIdx.x; e < sl; e += kThreads) {
    float g = 0.f;
    if (t == 1) {
      for (int i = beg; i < end; ++i) g += pc[(size_t)i * sl + e];
    } else {
      for (int i = beg; i < end; ++i) g += pc[(size_t)list[i] * sl + e];
    }
    if (optim == TTX_OPTIM_DENSE) {
      DW.c[t][base + e] = g;
    } else if (optim == TTX_OPTIM_SGD) {
      W.c[t][base + e] -= lr * g;
    } else {
      const float st = St.c[t][base + e] + g * g;
      St.c[t][base + e] = st;
      W.c[t][base + e] -= lr * g / (sqrtf(st) + eps);
    }
  }
}

// ---------------------------------------------------------- host side ------

static int check_lds(const Dims& d, const Lds& L) {
  if (L.MC <= 0 || L.bytes > 160 * 1024)
    TTX_FAIL(TTX_EUNSUPPORTED,
             "TT shape needs %d B of LDS per work-group (core-1 slice %d x %d floats); limit 163840",
             L.bytes, d.k[0], d.n[0]);
  return TTX_OK;
}

template <typename K>
static int allow_lds(K kernel, int bytes) {
  if (bytes > 64 * 1024) {
    TTX_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  }
  return TTX_OK;
}

static size_t rows_bytes(const Dims& d, long long nnz) { return align_up((size_t)nnz * d.D * 4); }

static int run_rows(const Dims& d, long long nnz, const Plan& P, const float* const* cores,
                    float* rows, hipStream_t st) {
  Lds L = make_lds(d, P.MC, false);
  int rc = check_lds(d, L);
  if (rc) return rc;
  rc = allow_lds(fwd_kernel, L.bytes);
  if (rc) return rc;
  CorePtrs C;
  for (int t = 0; t < TTX_MAX_CORES; ++t) C.c[t] = t < d.T ? (float*)cores[t] : nullptr;
  ProfScope ps(TTX_PROF_FWD, st);
  hipLaunchKernelGGL(fwd_kernel, dim3(P.max_chunks), dim3(kThreads), L.bytes, st, d, P, C, rows, L);
  TTX_HIP(hipGetLastError());
  return TTX_OK;
}

}  // namespace ttx

using namespace ttx;

extern "C" {

int ttx_set_chunk(int32_t mc) {
  if (mc < 0 || mc > 64) TTX_FAIL(TTX_EINVAL, "chunk %d out of range 0..64", mc);
  g_chunk_override = mc;
  return TTX_OK;
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
  if (choose_chunk(d) <= 0)
    TTX_FAIL(TTX_EUNSUPPORTED, "core-1 slice (%d x %d floats) does not fit the LDS tiling", d.k[0], d.n[0]);
  return TTX_OK;
}

int ttx_tt_forward(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz, const int64_t* indices,
                   const int64_t* rowidx, const int64_t* tableidx, const float* const* tt_cores,
                   float* output, const void* plan, void* workspace, size_t workspace_bytes,
                   ttx_stream_t stream) {
  Dims d;
  int rc = make_dims(g, &d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (B < 0 || !output) TTX_FAIL(TTX_EINVAL, "bad B/output");
  if ((size_t)d.num_tables * B * d.D > 0)
    TTX_HIP(hipMemsetAsync(output, 0, (size_t)d.num_tables * B * d.D * sizeof(float), st));
  if (nnz == 0) return TTX_OK;  // zeros, like cu:981-985
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
    rc = plan_build(d, nnz, indices, tableidx, P, st);
    if (rc) return rc;
    ws += pb;
  }
  float* rows = (float*)ws;
  rc = run_rows(d, nnz, P, tt_cores, rows, st);
  if (rc) return rc;
  {
    ProfScope ps(TTX_PROF_POOL, st);
    const int groups = kThreads / 32;
    hipLaunchKernelGGL(pool_kernel, dim3(((int)nnz + groups - 1) / groups), dim3(kThreads), 0, st,
                       (int)nnz, B, d.D, rowidx, tableidx, rows, output);
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
  rc = plan_build(d, nnz, indices, tableidx, P, (hipStream_t)stream);
  if (rc) return rc;
  return run_rows(d, nnz, P, tt_cores, rows, (hipStream_t)stream);
}

static size_t partial_bytes(const Dims& d, long long nnz, int MC, size_t* offs) {
  size_t o = 0;
  for (int t = 0; t < d.T; ++t) {
    offs[t] = o;
    const size_t cnt = (t == 1) ? (size_t)max_chunks(d, nnz, MC) : (size_t)nnz;
    o += align_up(cnt * d.slice[t] * sizeof(float));
  }
  return o;
}

size_t ttx_tt_backward_workspace_bytes(const ttx_geom* g, int32_t B, int32_t D, int64_t nnz) {
  (void)B; (void)D;
  Dims d;
  if (make_dims(g, &d) != TTX_OK || nnz < 0) return 0;
  const int MC = choose_chunk(d);
  if (MC <= 0) return 0;
  size_t offs[TTX_MAX_CORES];
  return plan_bytes(d, nnz) + partial_bytes(d, nnz, MC, offs) + 256;
}

int ttx_tt_backward(const ttx_geom* g, int32_t optim, int32_t B, int32_t D, float lr, float eps,
                    int64_t nnz, const int64_t* indices, const int64_t* rowidx,
                    const int64_t* tableidx, const float* d_output, float* const* tt_cores,
                    float* const* optimizer_state, float* const* d_tt_cores, const void* plan,
                    void* workspace, size_t workspace_bytes, ttx_stream_t stream) {
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
  const int MC = choose_chunk(d);
  size_t offs[TTX_MAX_CORES];
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
    rc = plan_build(d, nnz, indices, tableidx, P, st);
    if (rc) return rc;
    ws += pb;
  }
  Partials PC;
  for (int t = 0; t < TTX_MAX_CORES; ++t) PC.pc[t] = t < d.T ? (float*)(ws + offs[t]) : nullptr;
  Lds L = make_lds(d, P.MC, true);
  rc = check_lds(d, L);
  if (rc) return rc;
  rc = allow_lds(bwd_kernel, L.bytes);
  if (rc) return rc;
  CorePtrs C, S, DW;
  for (int t = 0; t < TTX_MAX_CORES; ++t) {
    C.c[t] = t < d.T ? tt_cores[t] : nullptr;
    S.c[t] = (t < d.T && optim == TTX_OPTIM_ADAGRAD) ? optimizer_state[t] : nullptr;
    DW.c[t] = (t < d.T && optim == TTX_OPTIM_DENSE) ? d_tt_cores[t] : nullptr;
  }
  {
    ProfScope ps(TTX_PROF_BWD, st);
    hipLaunchKernelGGL(bwd_kernel, dim3(P.max_chunks), dim3(kThreads), L.bytes, st, d, P, C, B,
                       rowidx, d_output, PC, L);
    TTX_HIP(hipGetLastError());
  }
  {
    int blocks = 0;
    for (int t = 0; t < d.T; ++t) blocks += d.S[t];
    ProfScope ps(TTX_PROF_APPLY, st);
    hipLaunchKernelGGL(reduce_apply_kernel, dim3(blocks), dim3(kThreads), 0, st, d, P, PC, optim, lr,
                       eps, C, S, DW);
    TTX_HIP(hipGetLastError());
  }
  return TTX_OK;
}

}  // extern "C"
