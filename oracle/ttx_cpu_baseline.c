/*
 * ttx_cpu_baseline.c -- the CPU restatement on ALL host cores, for bench.py's cpu_baseline leg ONLY.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE (same rules as ttx_oracle.c, which this file includes verbatim): never
 * linked into or called from the product.  The parity checker stays the scalar, sequential ttx_oracle.c; this
 * build exists because SURVEY.md section 8(d) asks for the CPU baseline "on all host cores":
 *
 *   forward : lookups are independent -> `omp parallel for` over the lookups' GEMM chains (rows kept in a
 *             [nnz, D] buffer), then `omp parallel for` over the bags, each summing its rows in index order
 *             (the same order as the sequential oracle -> bit-identical output);
 *   backward: `omp parallel for` over the lookups with one private gradient buffer per thread (the CPU
 *             counterpart of the reference's atomicAdd scatter, cu:362-377).  A thread zeroes a core slice of
 *             its buffer when it first touches it (with 128+ threads, zeroing and re-reading whole buffers
 *             costs several times the arithmetic); then every core slice is owned by one thread, which sums
 *             the buffers of the threads that touched it, in thread order, and applies the fused SGD /
 *             Adagrad update (untouched slices have g = 0: unchanged).
 *
 * Built by `make -C oracle baseline` with -O3 -fopenmp -mavx2 -mfma (x86-64-v3: any EPYC host of an MI355X).
 */
#include <omp.h>

#include "ttx_oracle.c"

int ttxo_omp_threads(void) { return omp_get_max_threads(); }

/* one fwd + fused-optimizer bwd step of a batch (what tt_embeddings_benchmark.py:183-187 times), all cores.
 * rows_ws: float[nnz * D]; grad_ws: float[threads * sum_t core elements] (need not be zeroed). */
int ttxo_omp_step(const ttx_geom* g, int32_t optim, int32_t B, int32_t D, float lr, float eps, int64_t nnz,
                  const int64_t* indices, const int64_t* offsets, const int64_t* rowidx, const int64_t* tableidx,
                  const float* d_output, float* const* cores, float* const* state, float* output, float* rows_ws,
                  float* grad_ws) {
  dims_t d;
  if (make_dims(g, &d)) return TTX_EINVAL;
  const int T = d.T;
  const int nthr = omp_get_max_threads();
  int64_t csz[TTX_MAX_CORES], coff[TTX_MAX_CORES + 1];
  coff[0] = 0;
  for (int t = 0; t < T; ++t) {
    csz[t] = (int64_t)g->num_tables * g->p[t] * d.slice[t];
    coff[t + 1] = coff[t] + csz[t];
  }
  const int64_t gtot = coff[T];
  int64_t maxs = d.max_x;
  for (int t = 0; t < T; ++t) if (d.slice[t] > maxs) maxs = d.slice[t];
  const int64_t nbags = (int64_t)g->num_tables * B;
  int64_t soff[TTX_MAX_CORES + 1]; /* slice numbering over all cores */
  soff[0] = 0;
  for (int t = 0; t < T; ++t) soff[t + 1] = soff[t] + (int64_t)g->num_tables * g->p[t];
  const int64_t stot = soff[T];
  unsigned char* touched = (unsigned char*)calloc((size_t)nthr * stot, 1);
  if (!touched) return TTX_EINVAL;
#pragma omp parallel
  {
    const int me = omp_get_thread_num();
    float* x[TTX_MAX_CORES];
    for (int t = 0; t < T - 1; ++t) x[t] = (float*)malloc(sizeof(float) * d.max_x);
    float* G = (float*)malloc(sizeof(float) * maxs);
    float* G2 = (float*)malloc(sizeof(float) * maxs);
    float* tmp = (float*)malloc(sizeof(float) * maxs);
    int64_t ii[TTX_MAX_CORES];
    /* ---- forward: rows, then bag sums in index order ---- */
#pragma omp for schedule(static)
    for (int64_t n = 0; n < nnz; ++n) {
      decode(&d, indices[n], ii);
      chain(g, &d, (const float* const*)cores, tableidx[n], ii, x);
      memcpy(rows_ws + n * D, x[T - 2], sizeof(float) * D);
    }
#pragma omp for schedule(static)
    for (int64_t b = 0; b < nbags; ++b) {
      float* o = output + b * D;
      for (int e = 0; e < D; ++e) o[e] = 0.0f;
      for (int64_t n = offsets[b]; n < offsets[b + 1]; ++n)
        for (int e = 0; e < D; ++e) o[e] += rows_ws[n * D + e];
    }
    /* ---- backward: private gradient buffers ---- */
    float* mine = grad_ws + (int64_t)me * gtot;
    unsigned char* mt = touched + (int64_t)me * stot;
#pragma omp for schedule(static)
    for (int64_t n = 0; n < nnz; ++n) {
      decode(&d, indices[n], ii);
      const int64_t tb = tableidx[n];
      chain(g, &d, (const float* const*)cores, tb, ii, x);
      memcpy(G, d_output + (tb * B + rowidx[n]) * D, sizeof(float) * D);
      for (int t = T - 2; t >= 0; --t) {
        const float* in = (t == 0) ? core_slice(g, &d, (const float* const*)cores, 0, tb, ii[0]) : x[t - 1];
        const float* ct = core_slice(g, &d, (const float* const*)cores, t + 1, tb, ii[t + 1]);
        gemm_tn(d.m[t], d.n[t], d.k[t], in, G, tmp);
        const int64_t sl = tb * g->p[t + 1] + ii[t + 1];
        float* dst = mine + coff[t + 1] + sl * d.slice[t + 1];
        if (!mt[soff[t + 1] + sl]) { mt[soff[t + 1] + sl] = 1; memset(dst, 0, sizeof(float) * d.slice[t + 1]); }
        for (int64_t e = 0; e < d.slice[t + 1]; ++e) dst[e] += tmp[e];
        gemm_nt(d.m[t], d.n[t], d.k[t], G, ct, G2);
        float* sw = G; G = G2; G2 = sw;
      }
      const int64_t sl0 = tb * g->p[0] + ii[0];
      float* dst0 = mine + coff[0] + sl0 * d.slice[0];
      if (!mt[soff[0] + sl0]) { mt[soff[0] + sl0] = 1; memset(dst0, 0, sizeof(float) * d.slice[0]); }
      for (int64_t e = 0; e < d.slice[0]; ++e) dst0[e] += G[e];
    }
    /* (implicit barrier) every slice has one owner: sum the touching threads' buffers in thread order, apply */
    for (int t = 0; t < T; ++t) {
      const int64_t ns = (int64_t)g->num_tables * g->p[t], ssz = d.slice[t];
#pragma omp for schedule(dynamic, 4)
      for (int64_t sl = 0; sl < ns; ++sl) {
        float* gsum = tmp; /* (slice <= maxs floats) */
        int any = 0;
        for (int k = 0; k < nthr; ++k) {
          if (!touched[(int64_t)k * stot + soff[t] + sl]) continue;
          const float* src = grad_ws + (int64_t)k * gtot + coff[t] + sl * ssz;
          if (!any) { memcpy(gsum, src, sizeof(float) * ssz); any = 1; }
          else for (int64_t e = 0; e < ssz; ++e) gsum[e] += src[e];
        }
        if (!any) continue;
        float* w = cores[t] + sl * ssz;
        if (optim == TTX_OPTIM_SGD) {
          for (int64_t e = 0; e < ssz; ++e) w[e] -= lr * gsum[e];
        } else if (optim == TTX_OPTIM_ADAGRAD) {
          float* st = state[t] + sl * ssz;
          for (int64_t e = 0; e < ssz; ++e) {
            const float gg = gsum[e];
            if (gg == 0.0f) continue;
            st[e] += gg * gg;
            w[e] -= lr * gg / (sqrtf(st[e]) + eps);
          }
        }
      }
    }
    for (int t = 0; t < T - 1; ++t) free(x[t]);
    free(G); free(G2); free(tmp);
  }
  free(touched);
  return TTX_OK;
}
