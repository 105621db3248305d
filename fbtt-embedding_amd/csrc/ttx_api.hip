// ttx_api.hip -- error reporting, geometry validation, live kernel timing.
#include <stdarg.h>

#include <map>
#include <mutex>
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

// device copy of a mixed geometry's per-table factors, created at first sight of the geometry
static const TabGeom* tab_geom_for(const ttx_geom* g) {
  static std::mutex mu;
  static std::map<std::vector<int>, const TabGeom*> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::vector<int> key = {dev, g->T, g->num_tables};
  key.insert(key.end(), g->p_tables, g->p_tables + (size_t)g->num_tables * g->T);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  std::vector<TabGeom> h(1);
  memset(h.data(), 0, sizeof(TabGeom));
  int base[TTX_MAX_CORES] = {0, 0, 0, 0};
  for (int k = 0; k < g->num_tables; ++k) {
    long long Lv = 1;
    for (int t = g->T - 1; t >= 0; --t) {
      h[0].p[k][t] = g->p_tables[(size_t)k * g->T + t];
      h[0].L[k][t] = Lv;
      Lv *= h[0].p[k][t];
    }
    for (int t = 0; t < g->T; ++t) { h[0].base[k][t] = base[t]; base[t] += h[0].p[k][t]; }
  }
  TabGeom* dptr = nullptr;
  if (hipMalloc((void**)&dptr, sizeof(TabGeom)) != hipSuccess) return nullptr;
  if (hipMemcpy(dptr, h.data(), sizeof(TabGeom), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  cache[key] = dptr;
  return dptr;
}

int allow_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;  // (kernel, device) -> bytes granted
  int dev = 0;
  TTX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  int& granted = done[std::make_pair(kernel, dev)];
  if (granted >= bytes) return TTX_OK;
  TTX_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  granted = bytes;
  return TTX_OK;
}

// compute units of the current device (persistent launches: two work-groups per CU); 256 if the runtime does not say
int device_cus() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lock(mu);
  int& n = cus[dev];
  if (n == 0 && (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)) n = 256;
  return n;
}

int make_dims(const ttx_geom* g, Dims* d) {
  // every entry point starts here: whatever error an earlier, unrelated runtime call left behind on this thread
  // is not this call's (the launches below check hipGetLastError() after themselves)
  (void)hipGetLastError();
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
  const bool mixed = g->p_tables && g->num_tables > 1;
  for (int t = g->T - 1; t >= 0; --t) {
    const int pt = mixed ? 1 : g->p[t];  // (mixed: p, S, L are set from p_tables below)
    if (pt <= 0 || g->q[t] <= 0 || g->r[t] <= 0)
      TTX_FAIL(TTX_EINVAL, "core %d: p, q, r must be > 0", t);
    d->p[t] = pt;
    d->q[t] = g->q[t];
    d->L[t] = Lv;
    Lv *= pt;
    Dv *= g->q[t];
    long long sl = (long long)g->r[t] * g->q[t] * g->r[t + 1];
    long long S = (long long)g->num_tables * pt;
    if (sl > (1ll << 30) || S > (1ll << 30) || Lv > (1ll << 62))
      TTX_FAIL(TTX_EINVAL, "core %d too large (slice %lld, slices %lld)", t, sl, S);
    d->slice[t] = (int)sl;
    d->S[t] = (int)S;
  }
  for (int t = 0; t <= g->T; ++t) d->r[t] = g->r[t];
  if (mixed) {  // tables of different row factors: S = the sums, p = the largest
    if (g->num_tables > TTX_MAX_TABLES_MIXED)
      TTX_FAIL(TTX_EINVAL, "num_tables=%d: at most %d tables with per-table row factors", g->num_tables,
               TTX_MAX_TABLES_MIXED);
    for (int t = 0; t < g->T; ++t) {
      long long S = 0;
      int pmax = 0;
      for (int k = 0; k < g->num_tables; ++k) {
        const int pk = g->p_tables[(size_t)k * g->T + t];
        if (pk <= 0) TTX_FAIL(TTX_EINVAL, "table %d core %d: p must be > 0", k, t);
        S += pk;
        pmax = pk > pmax ? pk : pmax;
      }
      if (S > (1ll << 30)) TTX_FAIL(TTX_EINVAL, "core %d: too many slices (%lld)", t, S);
      d->S[t] = (int)S;
      d->p[t] = pmax;
    }
    for (int k = 0; k < g->num_tables; ++k) {
      long long Lk = 1;
      for (int t = 0; t < g->T; ++t) {
        Lk *= g->p_tables[(size_t)k * g->T + t];
        if (Lk > (1ll << 62)) TTX_FAIL(TTX_EINVAL, "table %d: prod(p) too large", k);
      }
    }
    d->tab = tab_geom_for(g);
    if (!d->tab) TTX_FAIL(TTX_EHIP, "could not place the per-table geometry on the device");
    Lv = 1ll << 40;  // (no 32-bit decode)
  }
  d->idx32 = Lv <= (1ll << 32);
  for (int t = 0; t < g->T; ++t) {
    d->dvL[t] = make_udiv(d->idx32 ? (unsigned long long)d->L[t] : 1ull);
    d->dvP[t] = make_udiv((unsigned long long)d->p[t]);
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
// forward runs on the caller's thread, backward on the autograd engine's, MixedTTEmbeddingBag(streams=True) on
// several streams: the event lists are shared, so every access takes this lock.  (The `open` slot pairs ONE begin
// with the next end of the same kernel slot: with several streams in flight the pairs are still well-formed per
// thread because a launch's begin/end happen under one ProfScope on one thread; profile single-stream for timings
// that mean something.)
static std::mutex g_prof_mu;

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
  if (!(g_prof.mask >> which & 1u)) return;  // (unlocked fast path: profiling off)
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEvent_t e = get_event();
  if (!e) return;
  (void)hipEventRecord(e, s);
  g_prof.open[which] = e;
}

void prof_end(int which, hipStream_t s) {
  if (!g_prof.mask && !g_prof.open[which]) return;  // (unlocked fast path: profiling off, nothing open)
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof.open[which]) return;
  hipEvent_t e = get_event();
  if (!e) return;
  (void)hipEventRecord(e, s);
  g_prof.pending[which].push_back({g_prof.open[which], e});
  g_prof.open[which] = nullptr;
}

static void prof_drain() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
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
  // a failed event query (pairs recorded inside a captured graph that was not replayed since, ..) is not an
  // error of the next launch: do not leave it behind for that launch's hipGetLastError()
  (void)hipGetLastError();
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
  // Meant to be called right before a stream capture: the events the captured launches will record must exist
  // by then (creating one while a capture is open left an invalid handle behind on this stack -- the next launch
  // reported hipErrorInvalidResourceHandle), so the pool is stocked here.
  std::lock_guard<std::mutex> lk(ttx::g_prof_mu);
  if (mask) {
    while (ttx::g_prof.pool.size() < 512) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) break;
      ttx::g_prof.pool.push_back(e);
    }
  }
  ttx::g_prof.mask = (unsigned)mask;
  return TTX_OK;
}

int ttx_profile_reset(void) {
  ttx::prof_drain();
  std::lock_guard<std::mutex> lk(ttx::g_prof_mu);
  for (int w = 0; w < TTX_PROF_NUM; ++w) {
    ttx::g_prof.launches[w] = 0;
    ttx::g_prof.ms[w] = 0.0;
  }
  return TTX_OK;
}

int ttx_profile_read(int which, int64_t* launches, double* total_ms) {
  if (which < 0 || which >= TTX_PROF_NUM) TTX_FAIL(TTX_EINVAL, "profile slot %d out of range", which);
  ttx::prof_drain();
  std::lock_guard<std::mutex> lk(ttx::g_prof_mu);
  if (launches) *launches = ttx::g_prof.launches[which];
  if (total_ms) *total_ms = ttx::g_prof.ms[which];
  return TTX_OK;
}

}  // extern "C"
