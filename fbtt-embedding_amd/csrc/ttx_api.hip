// ttx_api.hip -- error reporting, geometry validation, live kernel timing.
#include <stdarg.h>

#include <vector>

#include "ttx_internal.h"

namespace ttx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static UDiv make_udiv(unsigned long long dd) {
  UDiv v;
  const unsigned d = (unsigned)dd;
  v.d = d;
  int l = 0;
  while ((1ull << l) < dd) ++l;  // ceil(log2 d)
  v.m = (unsigned)(((1ull << 32) * ((1ull << l) - dd)) / dd + 1);
  v.sh1 = l < 1 ? l : 1;
  v.sh2 = l > 1 ? l - 1 : 0;
  return v;
}

int make_dims(const ttx_geom* g, Dims* d) {
  if (!g) TTX_FAIL(TTX_EINVAL, "geometry is NULL");
  if (g->T < 2 || g->T > TTX_MAX_CORES)
    TTX_FAIL(TTX_EINVAL, "T=%d: number of TT cores must be 2..4", g->T);
  if (g->num_tables <= 0) TTX_FAIL(TTX_EINVAL, "num_tables=%d must be > 0", g->num_tables);
  memset(d, 0, sizeof(*d));
  d->T = g->T;
  d->num_tables = g->num_tables;
  if (g->r[0] != 1 || g->r[g->T] != 1)
    TTX_FAIL(TTX_EINVAL, "padded ranks must start and end with 1 (got %d, %d)", g->r[0], g->r[g->T]);
  long long Lv = 1;
  long long Dv = 1;
  for (int t = g->T - 1; t >= 0; --t) {
    if (g->p[t] <= 0 || g->q[t] <= 0 || g->r[t] <= 0)
      TTX_FAIL(TTX_EINVAL, "core %d: p, q, r must be > 0", t);
    d->p[t] = g->p[t];
    d->q[t] = g->q[t];
    d->L[t] = Lv;
    Lv *= g->p[t];
    Dv *= g->q[t];
    long long sl = (long long)g->r[t] * g->q[t] * g->r[t + 1];
    long long S = (long long)g->num_tables * g->p[t];
    if (sl > (1ll << 30) || S > (1ll << 30) || Lv > (1ll << 62))
      TTX_FAIL(TTX_EINVAL, "core %d too large (slice %lld, slices %lld)", t, sl, S);
    d->slice[t] = (int)sl;
    d->S[t] = (int)S;
  }
  for (int t = 0; t <= g->T; ++t) d->r[t] = g->r[t];
  d->idx32 = Lv <= (1ll << 32);
  for (int t = 0; t < g->T; ++t) {
    d->dvL[t] = make_udiv(d->idx32 ? (unsigned long long)d->L[t] : 1ull);
    d->dvP[t] = make_udiv((unsigned long long)g->p[t]);
  }
  if (Dv > (1ll << 24)) TTX_FAIL(TTX_EINVAL, "embedding_dim %lld too large", Dv);
  d->D = (int)Dv;
  int m_ = g->q[0];
  for (int t = 0; t < g->T - 1; ++t) {
    d->m[t] = m_;
    d->k[t] = g->r[t + 1];
    d->n[t] = g->q[t + 1] * g->r[t + 2];
    m_ *= g->q[t + 1];
  }
  return TTX_OK;
}

// ------------------------------------------------------------ profiling ----
struct ProfState {
  unsigned mask = 0;  // bit w: time kernel slot w
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[TTX_PROF_NUM];
  std::vector<hipEvent_t> pool;
  hipEvent_t open[TTX_PROF_NUM] = {};
  long long launches[TTX_PROF_NUM] = {};
  double ms[TTX_PROF_NUM] = {};
};
static ProfState g_prof;

static hipEvent_t get_event() {
  if (!g_prof.pool.empty()) {
    hipEvent_t e = g_prof.pool.back();
    g_prof.pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void prof_begin(int which, hipStream_t s) {
  if (!(g_prof.mask >> which & 1u)) return;
  hipEvent_t e = get_event();
  if (!e) return;
  (void)hipEventRecord(e, s);
  g_prof.open[which] = e;
}

void prof_end(int which, hipStream_t s) {
  if (!g_prof.open[which]) return;
  hipEvent_t e = get_event();
  if (!e) return;
  (void)hipEventRecord(e, s);
  g_prof.pending[which].push_back({g_prof.open[which], e});
  g_prof.open[which] = nullptr;
}

static void prof_drain() {
  for (int w = 0; w < TTX_PROF_NUM; ++w) {
    for (auto& pr : g_prof.pending[w]) {
      float ms = 0.f;
      if (hipEventSynchronize(pr.second) == hipSuccess &&
          hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        g_prof.ms[w] += ms;
        g_prof.launches[w] += 1;
      }
      g_prof.pool.push_back(pr.first);
      g_prof.pool.push_back(pr.second);
    }
    g_prof.pending[w].clear();
  }
}

}  // namespace ttx

extern "C" {

const char* ttx_last_error(void) { return ttx::g_err; }
int ttx_version(void) { return 100; }

int ttx_profile_enable(int mask) {
  ttx::prof_drain();
  ttx::g_prof.mask = (unsigned)mask;
  return TTX_OK;
}

int ttx_profile_mask(int mask) {  // as ttx_profile_enable, without reading back the pending event pairs
  ttx::g_prof.mask = (unsigned)mask;
  return TTX_OK;
}

int ttx_profile_reset(void) {
  ttx::prof_drain();
  for (int w = 0; w < TTX_PROF_NUM; ++w) {
    ttx::g_prof.launches[w] = 0;
    ttx::g_prof.ms[w] = 0.0;
  }
  return TTX_OK;
}

int ttx_profile_read(int which, int64_t* launches, double* total_ms) {
  if (which < 0 || which >= TTX_PROF_NUM) TTX_FAIL(TTX_EINVAL, "profile slot %d out of range", which);
  ttx::prof_drain();
  if (launches) *launches = ttx::g_prof.launches[which];
  if (total_ms) *total_ms = ttx::g_prof.ms[which];
  return TTX_OK;
}

}  // extern "C"
