// ttx_torch.cpp -- native (C++) autograd node of one TT lookup while the cache is not live.
//
// The reference's binding layer is C++ too (tt_embeddings.cpp, pybind11 over at::Tensor); its
// Python autograd.Function (tt_embeddings_ops.py:130-356) then costs ~200 us of interpreter,
// ctypes and autograd-engine time per training step here, three times the GPU time of the step
// (DESIGN.md section 6).  This file is the same node written against the C ABI of libttx.so:
// forward = ttx_lookup_prologue + ttx_tt_forward, backward = ttx_tt_backward (fused SGD / Adagrad
// in place, or dense core gradients), one lookup plan shared by both.  No compute happens here:
// torch supplies device memory (caching allocator), the current HIP stream and the autograd graph.
// TTCachedLookupOp is the cache-live variant (one table): hash lookup + stable partition, contraction of
// the misses, gather of the hits -- with the split point kept on the device (the reference reads it back
// and synchronises, tt_embeddings_cuda.cu:1481-1488; here the kernels read it from HBM); backward = fused TT update + cache-row update (or dense gradients).
// tt_embeddings_ops.py keeps the reference-shaped Python route and is the fallback when this
// extension was not built.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>
#include <torch/csrc/autograd/python_cpp_function.h>

#include <rccl/rccl.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <mutex>
#include <vector>

#include "ttx.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check(int rc) { TORCH_CHECK(rc == TTX_OK, "tt_embeddings (libttx): ", ttx_last_error()); }

// `p` holds T row factors -- or num_tables * T of them, table after table: tables of different row factors
// (include/ttx.h ttx_geom::p_tables).  The geometry then points into `ptab`, which must outlive its use.
struct Geom {
  ttx_geom g{};
  std::vector<int32_t> ptab;
};

void make_geom(Geom& G, int64_t num_tables, const std::vector<int64_t>& p, const std::vector<int64_t>& q,
               const std::vector<int64_t>& r) {
  const size_t T = q.size();
  const bool mixed = num_tables > 1 && p.size() == (size_t)num_tables * T && p.size() != T;
  TORCH_CHECK(T >= 2 && T <= TTX_MAX_CORES && (p.size() == T || mixed) && r.size() == T + 1,
              "tt_embeddings: need 2..4 cores with len(q) == len(p) and len(ranks) == len(p)+1");
  ttx_geom& g = G.g;
  g = ttx_geom{};
  g.T = (int32_t)T;
  g.num_tables = (int32_t)num_tables;
  for (size_t t = 0; t < T; ++t) { g.p[t] = mixed ? 0 : (int32_t)p[t]; g.q[t] = (int32_t)q[t]; }
  for (size_t t = 0; t <= T; ++t) g.r[t] = (int32_t)r[t];
  if (mixed) {
    G.ptab.assign(p.begin(), p.end());
    g.p_tables = G.ptab.data();
  }
}

void check_cores(const ttx_geom& g, at::TensorList cores, const char* what) {
  TORCH_CHECK((int64_t)cores.size() == g.T, "tt_embeddings: expected ", g.T, " ", what);
  for (int t = 0; t < g.T; ++t) {
    const Tensor& c = cores[t];
    int64_t n0 = g.num_tables, n1 = g.p[t];
    if (g.p_tables) {  // one array of all the tables' slices
      n0 = 1;
      n1 = 0;
      for (int k = 0; k < g.num_tables; ++k) n1 += g.p_tables[(size_t)k * g.T + t];
    }
    TORCH_CHECK(c.is_cuda() && c.scalar_type() == at::kFloat && c.is_contiguous() && c.dim() == 3 &&
                    c.size(0) == n0 && c.size(1) == n1 && c.size(2) == (int64_t)g.r[t] * g.q[t] * g.r[t + 1],
                "tt_embeddings: ", what, "[", t, "] must be a contiguous float32 GPU tensor of shape [", n0, ", ", n1,
                ", ", (int64_t)g.r[t] * g.q[t] * g.r[t + 1], "]");
  }
}

Tensor bytes_on(const Tensor& like, size_t n) {
  return at::empty({(int64_t)(n ? n : 1)}, like.options().dtype(at::kByte));
}

// The arrival counters of pooling fused into the forward kernel (ttx_tt_forward_o): one int per lookup, all zero before
// and after every call -- so ONE zero-initialised array per (device, stream) serves every step (no memset per call).
Tensor arrive_zeros(const Tensor& like, int64_t n, hipStream_t stream) {
  static std::mutex mu;
  static std::map<std::pair<int, void*>, Tensor> cache;
  std::lock_guard<std::mutex> lock(mu);
  Tensor& t = cache[std::make_pair((int)like.get_device(), (void*)stream)];
  if (!t.defined() || t.numel() < n) t = at::zeros({std::max<int64_t>(n, 1 << 16)}, like.options().dtype(at::kInt));
  return t;
}

struct TTLookupOp : public torch::autograd::Function<TTLookupOp> {
  // inputs: indices, offsets, hashtbl, cache_freq (the last two may be undefined), then
  // T optimizer-state tensors (empty list unless Adagrad), then the T cores.
  static Tensor forward(AutogradContext* ctx, const Tensor& indices, const Tensor& offsets, int64_t num_tables,
                        std::vector<int64_t> p, std::vector<int64_t> q, std::vector<int64_t> r, int64_t optim,
                        double lr, double eps, const c10::optional<Tensor>& hashtbl,
                        const c10::optional<Tensor>& cache_freq, const c10::optional<Tensor>& psw,
                        const c10::optional<Tensor>& pre_rowidx, const c10::optional<Tensor>& pre_tableidx,
                        const c10::optional<Tensor>& pre_plan, at::TensorList state, at::TensorList cores) {
    Geom G;
    make_geom(G, num_tables, p, q, r);
    const ttx_geom& g = G.g;
    check_cores(g, cores, "tt_cores");
    const bool weighted = psw.has_value() && psw->defined();
    // the lookup prologue of this batch may have run ahead on another stream (`prologue` below; the module's prefetch)
    const bool pre = pre_plan.has_value() && pre_plan->defined();
    if (weighted)
      TORCH_CHECK(psw->is_cuda() && psw->scalar_type() == at::kFloat && psw->is_contiguous() &&
                      psw->numel() == indices.numel(),
                  "tt_embeddings: per_sample_weights must be a contiguous float32 GPU tensor, one weight per index");
    TORCH_CHECK(indices.is_cuda() && indices.scalar_type() == at::kLong && indices.is_contiguous() &&
                    offsets.is_cuda() && offsets.scalar_type() == at::kLong && offsets.is_contiguous(),
                "tt_embeddings: indices / offsets must be contiguous int64 GPU tensors");
    if (optim == TTX_OPTIM_ADAGRAD) check_cores(g, state, "optimizer_state");
    const int64_t nnz = indices.numel(), nb = offsets.numel() - 1;
    TORCH_CHECK(nb > 0 && nb % num_tables == 0, "tt_embeddings: offsets must hold num_tables * B + 1 entries");
    const int64_t B = nb / num_tables;
    int64_t D = 1;
    for (auto v : q) D *= v;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(indices.device());  // (torch-ROCm calls its HIP devices "cuda")
    auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();

    Tensor out = at::empty({num_tables, B, D}, cores[0].options());
    // bag rows, table ids and the lookup plan of this batch: the planned-ahead tensors, or ONE buffer
    // [rowidx | tableidx | plan] (the eager step is host-bound: every allocation saved is a microsecond)
    Tensor rowidx_t, tableidx_t, plan_t, buf;
    int64_t *rowidx_p = nullptr, *tableidx_p = nullptr;
    void* plan_p = nullptr;
    if (pre) {
      TORCH_CHECK(pre_rowidx.has_value() && pre_tableidx.has_value() && pre_rowidx->numel() == nnz &&
                      pre_tableidx->numel() == nnz && (size_t)pre_plan->numel() >= ttx_plan_bytes(&g, nnz),
                  "tt_embeddings: the prefetched prologue does not belong to this batch");
      rowidx_t = *pre_rowidx;
      tableidx_t = *pre_tableidx;
      plan_t = *pre_plan;
      rowidx_p = rowidx_t.data_ptr<int64_t>();
      tableidx_p = tableidx_t.data_ptr<int64_t>();
      plan_p = plan_t.data_ptr();
    } else {
      const size_t pb = nnz > 0 ? ttx_plan_bytes(&g, nnz) : 0;
      const size_t ib = ((size_t)nnz * 8 + 255) / 256 * 256;
      buf = bytes_on(indices, 2 * ib + pb);
      char* base = (char*)buf.data_ptr();
      rowidx_p = (int64_t*)base;
      tableidx_p = (int64_t*)(base + ib);
      plan_p = nnz > 0 ? (void*)(base + 2 * ib) : nullptr;
      if (nnz > 0) {
        const bool upd = hashtbl.has_value() && hashtbl->defined() && hashtbl->numel() > 0 && cache_freq.has_value() &&
                         cache_freq->defined();
        if (upd) TORCH_CHECK(hashtbl->numel() == cache_freq->numel(), "tt_embeddings: hashtbl must match cache_freq");
        check(ttx_lookup_prologue(&g, nnz, indices.data_ptr<int64_t>(), nb, offsets.data_ptr<int64_t>(),
                                  upd ? hashtbl->numel() : 0, upd ? hashtbl->data_ptr<int64_t>() : nullptr,
                                  upd ? cache_freq->data_ptr<int64_t>() : nullptr, rowidx_p, tableidx_p, plan_p, pb, stream));
      }
    }
    const float* cp[TTX_MAX_CORES] = {};
    for (int t = 0; t < g.T; ++t) cp[t] = cores[t].data_ptr<float>();
    const size_t wb = ttx_tt_forward_workspace_bytes(&g, (int32_t)B, (int32_t)D, nnz);
    Tensor ws = bytes_on(indices, wb);
    // a gradient for the weights needs the lookups' rows in backward: d_psw[n] = <d_out[bag(n)], row_n>
    const bool psw_grad = weighted && psw->requires_grad() && nnz > 0;
    Tensor rows_keep;
    if (psw_grad) rows_keep = at::empty({nnz, D}, cores[0].options());
    // bag pooling inside the contraction kernel where the shape allows it (the bags are named by `offsets`).  Opt-in
    // (TTX_FUSED_POOL=1): measured at the benchmark batch it saves the pooling launch (-5.3 us) and pays it back in the
    // forward kernel's tail (+5.7 us: the completing lookups' dependent reads of rows that have just been written
    // through to memory, clustered in the last work-groups to finish) -- DESIGN.md section 4.5.
    static const bool fuse_pool = std::getenv("TTX_FUSED_POOL") != nullptr && std::getenv("TTX_FUSED_POOL")[0] == '1';
    const int64_t na = (nnz > 0 && fuse_pool) ? ttx_tt_forward_arrive_ints(&g, nnz) : 0;
    Tensor arrive;
    if (na > 0) arrive = arrive_zeros(indices, na, stream);
    check(ttx_tt_forward_o(&g, (int32_t)B, (int32_t)D, nnz, indices.data_ptr<int64_t>(), rowidx_p, tableidx_p,
                           weighted ? psw->data_ptr<float>() : nullptr, cp,
                           out.data_ptr<float>(), psw_grad ? rows_keep.data_ptr<float>() : nullptr,
                           na > 0 ? offsets.data_ptr<int64_t>() : nullptr, na > 0 ? arrive.data_ptr<int32_t>() : nullptr,
                           nnz > 0 ? plan_p : nullptr, ws.data_ptr(), wb, stream));

    // What backward needs, in THREE saved entries (every saved_data entry is a string-keyed map insertion plus an IValue):
    //   "m": {num_tables, optim, T, nstate, prefetched, weighted, psw_grad, |p|, p.., q.., r..}   "d": {lr, eps}
    //   "t": indices, then [rowidx, tableidx, plan] (prefetched) or [buf], then cores.., state.., then psw, rows if present
    // (integer tensors and the in-place-updated cores / state are kept out of the version-counter check on purpose: the
    //  fused optimizer mutates the cores between forward and the next backward)
    std::vector<int64_t> meta = {num_tables, optim, (int64_t)g.T, (int64_t)state.size(), pre ? 1 : 0, weighted ? 1 : 0,
                                 psw_grad ? 1 : 0, (int64_t)p.size()};
    meta.insert(meta.end(), p.begin(), p.end());
    meta.insert(meta.end(), q.begin(), q.end());
    meta.insert(meta.end(), r.begin(), r.end());
    std::vector<Tensor> keep;
    keep.reserve(6 + cores.size() + state.size());
    keep.push_back(indices);
    if (pre) { keep.push_back(rowidx_t); keep.push_back(tableidx_t); keep.push_back(plan_t); }
    else keep.push_back(buf);
    keep.insert(keep.end(), cores.begin(), cores.end());
    keep.insert(keep.end(), state.begin(), state.end());
    if (weighted) keep.push_back(psw->detach());
    if (psw_grad) keep.push_back(rows_keep);
    ctx->saved_data["m"] = std::move(meta);
    ctx->saved_data["d"] = std::vector<double>{lr, eps};
    ctx->saved_data["t"] = std::move(keep);
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grad_outputs) {
    const auto meta = ctx->saved_data["m"].toIntVector();
    const auto lre = ctx->saved_data["d"].toDoubleVector();
    const auto keep = ctx->saved_data["t"].toTensorVector();
    const int64_t num_tables = meta[0], optim = meta[1], T = meta[2], nstate = meta[3];
    const bool pre = meta[4] != 0, weighted = meta[5] != 0, psw_grad = meta[6] != 0;
    const int64_t np_ = meta[7];
    const std::vector<int64_t> p(meta.begin() + 8, meta.begin() + 8 + np_);
    const std::vector<int64_t> q(meta.begin() + 8 + np_, meta.begin() + 8 + np_ + T);
    const std::vector<int64_t> r(meta.begin() + 8 + np_ + T, meta.begin() + 8 + np_ + 2 * T + 1);
    const double lr = lre[0], eps = lre[1];
    Geom G;
    make_geom(G, num_tables, p, q, r);
    const ttx_geom& g = G.g;
    const Tensor& indices = keep[0];
    const int64_t nnz = indices.numel();
    const int64_t *rowidx_p, *tableidx_p;
    const void* plan_p;
    size_t at = 1;
    if (pre) {
      rowidx_p = keep[1].data_ptr<int64_t>();
      tableidx_p = keep[2].data_ptr<int64_t>();
      plan_p = nnz > 0 ? keep[3].data_ptr() : nullptr;
      at = 4;
    } else {
      const size_t ib = ((size_t)nnz * 8 + 255) / 256 * 256;
      const char* base = (const char*)keep[1].data_ptr();
      rowidx_p = (const int64_t*)base;
      tableidx_p = (const int64_t*)(base + ib);
      plan_p = nnz > 0 ? (const void*)(base + 2 * ib) : nullptr;
      at = 2;
    }
    const Tensor* cores = &keep[at];
    const Tensor* state = &keep[at + T];
    const Tensor psw = weighted ? keep[at + T + nstate] : Tensor();

    // one slot per forward argument (lists expanded): indices, offsets, num_tables, p, q, r, optim, lr, eps,
    // hashtbl, cache_freq, per_sample_weights, pre_rowidx, pre_tableidx, pre_plan, state.., cores..
    constexpr int64_t kHead = 15;
    variable_list grads(kHead + nstate + T);
    Tensor go = grad_outputs[0];
    TORCH_CHECK(go.defined(), "tt_embeddings: backward needs the output gradient");
    go = go.contiguous();
    TORCH_CHECK(go.scalar_type() == at::kFloat && go.dim() == 3 && go.size(0) == num_tables,
                "tt_embeddings: d_output must be float32 [num_tables, B, D]");
    const int64_t B = go.size(1), D = go.size(2);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(go.device());
    auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();

    float* cp[TTX_MAX_CORES] = {};
    float* sp[TTX_MAX_CORES] = {};
    float* gp[TTX_MAX_CORES] = {};
    std::vector<Tensor> dense;
    for (int t = 0; t < T; ++t) {
      cp[t] = cores[t].data_ptr<float>();
      if (optim == TTX_OPTIM_ADAGRAD) sp[t] = state[t].data_ptr<float>();
      if (optim == TTX_OPTIM_DENSE) {
        dense.push_back(at::empty_like(cores[t]));
        gp[t] = dense.back().data_ptr<float>();
      }
    }
    const size_t wb = ttx_tt_backward_workspace_bytes(&g, (int32_t)B, (int32_t)D, nnz);
    Tensor ws = bytes_on(indices, wb);
    check(ttx_tt_backward_w(&g, (int32_t)optim, (int32_t)B, (int32_t)D, (float)lr, (float)eps, nnz,
                            indices.data_ptr<int64_t>(), rowidx_p, tableidx_p,
                            psw.defined() ? psw.data_ptr<float>() : nullptr, go.data_ptr<float>(), cp,
                            optim == TTX_OPTIM_ADAGRAD ? sp : nullptr, optim == TTX_OPTIM_DENSE ? gp : nullptr,
                            plan_p, ws.data_ptr(), wb, stream));
    if (optim == TTX_OPTIM_DENSE)
      for (int t = 0; t < T; ++t) grads[kHead + nstate + t] = dense[t];
    if (psw_grad) {  // gradient of the per_sample_weights (argument slot 11)
      const Tensor& rows = keep[at + T + nstate + 1];
      Tensor d_psw = at::empty({nnz}, rows.options());
      check(ttx_psw_backward((int32_t)B, (int32_t)D, nnz, rows.data_ptr<float>(), rowidx_p, tableidx_p, go.data_ptr<float>(),
                             d_psw.data_ptr<float>(), stream));
      grads[11] = d_psw;
    }
    return grads;
  }
};

Tensor lookup(const Tensor& indices, const Tensor& offsets, int64_t num_tables, std::vector<int64_t> p,
              std::vector<int64_t> q, std::vector<int64_t> r, int64_t optim, double lr, double eps,
              c10::optional<Tensor> hashtbl, c10::optional<Tensor> cache_freq, std::vector<Tensor> state,
              std::vector<Tensor> cores, c10::optional<Tensor> per_sample_weights, c10::optional<Tensor> pre_rowidx,
              c10::optional<Tensor> pre_tableidx, c10::optional<Tensor> pre_plan) {
  // (weights that require a gradient get one: d_psw[n] = <d_out[bag(n)], row_n>; the rows are kept for it)
  return TTLookupOp::apply(indices, offsets, num_tables, std::move(p), std::move(q), std::move(r), optim, lr, eps,
                           hashtbl, cache_freq, per_sample_weights, pre_rowidx, pre_tableidx, pre_plan,
                           at::TensorList(state), at::TensorList(cores));
}

// The lookup prologue alone (cache not live): frequency update, offsets -> bag rows, lookup plan -- everything of a
// step that depends on the batch's indices only, not on the cores.  Enqueued on the CURRENT stream: the module's
// prefetch runs it on a side stream while the previous step's backward still occupies the main one, and hands the
// three tensors to `lookup`.  -> {rowidx, tableidx, plan}
// n_dev (optional, int32 [1] on the device): the number of LIVE lookups -- indices holds an upper bound, offsets describes exactly
// n_dev of them; the plan (and with it every kernel of the lookup that takes these three tensors) covers only those.
std::vector<Tensor> prologue(const Tensor& indices, const Tensor& offsets, int64_t num_tables, std::vector<int64_t> p,
                             std::vector<int64_t> q, std::vector<int64_t> r, c10::optional<Tensor> hashtbl,
                             c10::optional<Tensor> cache_freq, c10::optional<Tensor> n_dev) {
  Geom G;
  make_geom(G, num_tables, p, q, r);
  const ttx_geom& g = G.g;
  TORCH_CHECK(indices.is_cuda() && indices.scalar_type() == at::kLong && indices.is_contiguous() && offsets.is_cuda() &&
                  offsets.scalar_type() == at::kLong && offsets.is_contiguous(),
              "tt_embeddings: indices / offsets must be contiguous int64 GPU tensors");
  const int64_t nnz = indices.numel(), nb = offsets.numel() - 1;
  TORCH_CHECK(nnz > 0 && nb > 0 && nb % num_tables == 0, "tt_embeddings: offsets must hold num_tables * B + 1 entries");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(indices.device());
  auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  Tensor rowidx = at::empty_like(indices), tableidx = at::empty_like(indices);
  const size_t pb = ttx_plan_bytes(&g, nnz);
  Tensor plan = bytes_on(indices, pb);
  const bool upd = hashtbl.has_value() && hashtbl->defined() && hashtbl->numel() > 0 && cache_freq.has_value() &&
                   cache_freq->defined();
  if (upd) TORCH_CHECK(hashtbl->numel() == cache_freq->numel(), "tt_embeddings: hashtbl must match cache_freq");
  const bool live = n_dev.has_value() && n_dev->defined();
  if (live)
    TORCH_CHECK(n_dev->is_cuda() && n_dev->scalar_type() == at::kInt && n_dev->numel() == 1,
                "tt_embeddings: n_dev must be one int32 on the GPU");
  check(ttx_lookup_prologue_n(&g, nnz, indices.data_ptr<int64_t>(), nb, offsets.data_ptr<int64_t>(),
                              upd ? hashtbl->numel() : 0, upd ? hashtbl->data_ptr<int64_t>() : nullptr,
                              upd ? cache_freq->data_ptr<int64_t>() : nullptr, rowidx.data_ptr<int64_t>(),
                              tableidx.data_ptr<int64_t>(), plan.data_ptr(), pb, live ? n_dev->data_ptr<int32_t>() : nullptr,
                              stream));
  return {rowidx, tableidx, plan};
}

// ... and of several batches at once (ttx_lookup_prologue_multi: one launch per 16 batches): every batch with the same
// number of indices and offsets.  -> {rowidx [nbatch, nnz], tableidx [nbatch, nnz], plans [nbatch, stride]}; row k of each is
// what `lookup(pre_*=)` takes for batch k.
std::vector<Tensor> prologue_multi(std::vector<Tensor> indices, std::vector<Tensor> offsets, int64_t num_tables,
                                   std::vector<int64_t> p, std::vector<int64_t> q, std::vector<int64_t> r,
                                   c10::optional<Tensor> hashtbl, c10::optional<Tensor> cache_freq) {
  Geom G;
  make_geom(G, num_tables, p, q, r);
  const ttx_geom& g = G.g;
  const int64_t nbatch = (int64_t)indices.size();
  TORCH_CHECK(nbatch > 0 && offsets.size() == indices.size(), "tt_embeddings: one offsets tensor per indices tensor");
  const int64_t nnz = indices[0].numel(), nb = offsets[0].numel() - 1;
  std::vector<const int64_t*> ip(nbatch), op(nbatch);
  for (int64_t k = 0; k < nbatch; ++k) {
    TORCH_CHECK(indices[k].is_cuda() && indices[k].scalar_type() == at::kLong && indices[k].is_contiguous() &&
                    offsets[k].is_cuda() && offsets[k].scalar_type() == at::kLong && offsets[k].is_contiguous() &&
                    indices[k].numel() == nnz && offsets[k].numel() == nb + 1,
                "tt_embeddings: the batches of a multi-batch prologue must be contiguous int64 GPU tensors of one size");
    ip[k] = indices[k].data_ptr<int64_t>();
    op[k] = offsets[k].data_ptr<int64_t>();
  }
  TORCH_CHECK(nnz > 0 && nb > 0 && nb % num_tables == 0, "tt_embeddings: offsets must hold num_tables * B + 1 entries");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(indices[0].device());
  auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  Tensor rowidx = at::empty({nbatch, nnz}, indices[0].options()), tableidx = at::empty({nbatch, nnz}, indices[0].options());
  const size_t stride = (ttx_plan_bytes(&g, nnz) + 255) / 256 * 256;
  Tensor plans = at::empty({nbatch, (int64_t)stride}, indices[0].options().dtype(at::kByte));
  const bool upd = hashtbl.has_value() && hashtbl->defined() && hashtbl->numel() > 0 && cache_freq.has_value() &&
                   cache_freq->defined();
  if (upd) TORCH_CHECK(hashtbl->numel() == cache_freq->numel(), "tt_embeddings: hashtbl must match cache_freq");
  check(ttx_lookup_prologue_multi(&g, (int32_t)nbatch, nnz, ip.data(), nb, op.data(), upd ? hashtbl->numel() : 0,
                                  upd ? hashtbl->data_ptr<int64_t>() : nullptr,
                                  upd ? cache_freq->data_ptr<int64_t>() : nullptr, rowidx.data_ptr<int64_t>(),
                                  tableidx.data_ptr<int64_t>(), plans.data_ptr(), stride, stream));
  return {rowidx, tableidx, plans};
}

// ---- cache live (one table): tt_embeddings_ops.py:821-874 with self.warmup == False ----------------
struct TTCachedLookupOp : public torch::autograd::Function<TTCachedLookupOp> {
  // args: indices, offsets, p, q, r, optim, lr, eps, hashtbl, cache_freq, cache_state,
  //       cache_optimizer_state (undefined unless Adagrad), cache_weight, per_sample_weights (optional),
  //       pre (the batch's row of prologue_cached_multi's results: {tableidx, pcol, prow, ploc, n_tt, plan}, or empty),
  //       state.., cores..
  static constexpr int64_t kHead = 14;  // (then one slot per tensor of pre, state, cores)
  static Tensor forward(AutogradContext* ctx, const Tensor& indices, const Tensor& offsets, std::vector<int64_t> p,
                        std::vector<int64_t> q, std::vector<int64_t> r, int64_t optim, double lr, double eps,
                        const Tensor& hashtbl, const Tensor& cache_freq, const Tensor& cache_state,
                        const c10::optional<Tensor>& cache_opt_state, const Tensor& cache_weight,
                        const c10::optional<Tensor>& psw, at::TensorList pre, at::TensorList state,
                        at::TensorList cores) {
    // bit 8 of `optim`: this batch's frequency update has been issued already (a planned-ahead prologue that was then
    // discarded, tt_embeddings_ops.py): look the indices up without counting them a second time
    const bool count_freq = (optim & 256) == 0;
    // bits 9 / 10 of `optim`: the cache rows' update of the backward -- 9: sorted, atomic-free, bit-identical from run to run
    // (ttx_cache_backward_sorted); 10: the one-launch float-atomic kernels; neither: sorted from kSortedAutoNnz lookups on
    const int64_t det = (optim >> 9) & 3;
    optim &= 255;
    Geom G;
    make_geom(G, 1, p, q, r);
    const ttx_geom& g = G.g;
    check_cores(g, cores, "tt_cores");
    if (optim == TTX_OPTIM_ADAGRAD) check_cores(g, state, "optimizer_state");
    TORCH_CHECK(indices.is_cuda() && indices.scalar_type() == at::kLong && indices.is_contiguous() &&
                    offsets.is_cuda() && offsets.scalar_type() == at::kLong && offsets.is_contiguous(),
                "tt_embeddings: indices / offsets must be contiguous int64 GPU tensors");
    const int64_t H = hashtbl.numel();
    TORCH_CHECK(H > 0 && cache_freq.numel() == H && cache_state.numel() == H && hashtbl.scalar_type() == at::kLong &&
                    cache_freq.scalar_type() == at::kLong && cache_state.scalar_type() == at::kInt,
                "tt_embeddings: hashtbl / cache_freq (int64) and cache_state (int32) must have hashtbl_size entries");
    TORCH_CHECK(cache_weight.is_cuda() && cache_weight.scalar_type() == at::kFloat && cache_weight.is_contiguous() &&
                    cache_weight.dim() == 2, "tt_embeddings: cache_weight must be contiguous float32 [cache_size, D]");
    const int64_t nnz = indices.numel(), B = offsets.numel() - 1, D = cache_weight.size(1);
    TORCH_CHECK(B > 0 && nnz > 0, "tt_embeddings: empty batch");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(indices.device());
    auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();

    const bool weighted = psw.has_value() && psw->defined();
    if (weighted)
      TORCH_CHECK(psw->is_cuda() && psw->scalar_type() == at::kFloat && psw->is_contiguous() && psw->numel() == nnz &&
                      pre.size() == 0,
                  "tt_embeddings: per_sample_weights must be a contiguous float32 GPU tensor, one weight per index "
                  "(and the batch's prologue runs in line)");
    const bool psw_grad = weighted && psw->requires_grad();
    Tensor ppsw, porig, buf;
    std::vector<Tensor> pre_t;
    // the partitioned arrays, the split point and the plan of the misses: the six planned-ahead tensors, or ONE buffer
    // [tableidx | pcol | prow | ploc | n_tt | plan | rowidx, preprocessing scratch] (host-bound step: one allocation, not eight)
    int64_t *tableidx_p, *pcol_p, *prow_p;
    int32_t *ploc_p, *ntt_p;
    void* plan_p;
    const size_t pb = ttx_plan_bytes(&g, nnz);
    const size_t ib = ((size_t)nnz * 8 + 255) / 256 * 256, lb = ((size_t)nnz * 4 + 255) / 256 * 256;
    if (pre.size() == 6) {  // planned ahead (prologue_cached_multi): frequency update, lookup, partition and plan are done
      const Tensor &tableidx = pre[0], &pcol = pre[1], &prow = pre[2], &ploc = pre[3], &n_tt = pre[4], &plan = pre[5];
      TORCH_CHECK(tableidx.numel() == nnz && pcol.numel() == nnz && prow.numel() == nnz && ploc.numel() == nnz &&
                      n_tt.numel() == 1 && (size_t)plan.numel() >= pb && pcol.scalar_type() == at::kLong &&
                      ploc.scalar_type() == at::kInt && n_tt.scalar_type() == at::kInt && plan.is_contiguous(),
                  "tt_embeddings: the planned-ahead prologue does not match this batch");
      pre_t.assign(pre.begin(), pre.end());
      tableidx_p = tableidx.data_ptr<int64_t>(); pcol_p = pcol.data_ptr<int64_t>(); prow_p = prow.data_ptr<int64_t>();
      ploc_p = ploc.data_ptr<int32_t>(); ntt_p = n_tt.data_ptr<int32_t>(); plan_p = plan.data_ptr();
    } else {
      TORCH_CHECK(pre.size() == 0, "tt_embeddings: pre must be prologue_cached_multi's six tensors of the batch, or empty");
      const size_t pwb = ttx_preprocess_workspace_bytes(nnz);
      const size_t pbA = (pb + 255) / 256 * 256;
      buf = bytes_on(indices, 3 * ib + lb + 256 + pbA + ib + pwb);
      char* base = (char*)buf.data_ptr();
      tableidx_p = (int64_t*)base; pcol_p = (int64_t*)(base + ib); prow_p = (int64_t*)(base + 2 * ib);
      ploc_p = (int32_t*)(base + 3 * ib); ntt_p = (int32_t*)(base + 3 * ib + lb); plan_p = base + 3 * ib + lb + 256;
      int64_t* rowidx_p = (int64_t*)(base + 3 * ib + lb + 256 + pbA);  // (scratch: only the partitioned rows are used later)
      void* pws_p = base + 3 * ib + lb + 256 + pbA + ib;
      // the split point (number of TT entries) stays on the device: the kernels below read it there,
      // nnz only sizes grids and workspaces -- no host synchronisation, the step can be graph-captured
      int32_t n_host = 0, part = 0;
      if (weighted) {  // the weights follow their lookups through the partition; the origins bring their gradient back
        ppsw = at::empty({nnz}, cores[0].options());
        if (psw_grad) porig = at::empty({nnz}, indices.options().dtype(at::kInt));
      }
      check(ttx_preprocess_indices_async_w(nnz, indices.data_ptr<int64_t>(), B, offsets.data_ptr<int64_t>(), 1, 0, H,
                                           hashtbl.data_ptr<int64_t>(), cache_state.data_ptr<int32_t>(),
                                           rowidx_p, tableidx_p, pcol_p, prow_p, ploc_p,
                                           &n_host, &part, ntt_p,
                                           count_freq ? hashtbl.data_ptr<int64_t>() : nullptr,
                                           count_freq ? cache_freq.data_ptr<int64_t>() : nullptr, weighted ? psw->data_ptr<float>() : nullptr,
                                           weighted ? ppsw.data_ptr<float>() : nullptr,
                                           psw_grad ? porig.data_ptr<int32_t>() : nullptr, pws_p, pwb, stream));
      TORCH_CHECK(part == 1, "tt_embeddings: the cache-live preprocessing did not partition");
      check(ttx_plan_build_n(&g, nnz, ntt_p, pcol_p, tableidx_p, prow_p, plan_p, pb, stream));
    }

    Tensor out = at::empty({1, B, D}, cores[0].options());
    const float* cp[TTX_MAX_CORES] = {};
    for (int t = 0; t < g.T; ++t) cp[t] = cores[t].data_ptr<float>();
    const size_t wb = ttx_tt_forward_workspace_bytes(&g, (int32_t)B, (int32_t)D, nnz);
    Tensor ws = bytes_on(indices, wb);
    // (a gradient for the weights needs every lookup's forward row: the contraction's for the misses, the cache's for the hits)
    Tensor rows_keep;
    if (psw_grad) rows_keep = at::empty({nnz, D}, cores[0].options());
    if (!weighted && ttx_tt_forward_cached_supported(&g, (int32_t)D, nnz)) {
      // contraction of the misses, then ONE launch for the bag sums of both parts (pooling + cache gather)
      check(ttx_tt_forward_cached(&g, (int32_t)B, (int32_t)D, nnz, pcol_p, prow_p, tableidx_p, cp, ploc_p,
                                  cache_weight.data_ptr<float>(), out.data_ptr<float>(), plan_p, ws.data_ptr(), wb, stream));
    } else {
      check(ttx_tt_forward_wr(&g, (int32_t)B, (int32_t)D, nnz, pcol_p, prow_p, tableidx_p,
                              weighted ? ppsw.data_ptr<float>() : nullptr, cp,
                              out.data_ptr<float>(), psw_grad ? rows_keep.data_ptr<float>() : nullptr, plan_p,
                              ws.data_ptr(), wb, stream));
      check(ttx_cache_forward_nw((int32_t)B, nnz, ntt_p, ploc_p, prow_p, weighted ? ppsw.data_ptr<float>() : nullptr, (int32_t)D,
                                 cache_weight.data_ptr<float>(), out.data_ptr<float>(), stream));
    }
    if (psw_grad)
      check(ttx_cache_rows_n(nnz, ntt_p, ploc_p, (int32_t)D, cache_weight.data_ptr<float>(), rows_keep.data_ptr<float>(), stream));

    // three saved entries (see TTLookupOp::forward): "m" = {optim, T, nstate, npre, weighted, psw_grad, has_copt, has_plan, p.., q.., r..},
    // "d" = {lr, eps}, "t" = cache_weight, then the six planned-ahead tensors or [buf], [copt], cores.., state.., [ppsw], [rows, porig]
    const bool has_copt = cache_opt_state.has_value() && cache_opt_state->defined();
    std::vector<int64_t> meta = {optim | (det << 9), (int64_t)g.T, (int64_t)state.size(), (int64_t)pre.size(), weighted ? 1 : 0,
                                 psw_grad ? 1 : 0, has_copt ? 1 : 0, nnz};
    meta.insert(meta.end(), p.begin(), p.end());
    meta.insert(meta.end(), q.begin(), q.end());
    meta.insert(meta.end(), r.begin(), r.end());
    std::vector<Tensor> keep = {cache_weight};
    keep.reserve(14 + cores.size() + state.size());
    if (pre.size() == 6) keep.insert(keep.end(), pre_t.begin(), pre_t.end());
    else keep.push_back(buf);
    if (has_copt) keep.push_back(*cache_opt_state);
    keep.insert(keep.end(), cores.begin(), cores.end());
    keep.insert(keep.end(), state.begin(), state.end());
    if (weighted) keep.push_back(ppsw);
    if (psw_grad) { keep.push_back(rows_keep); keep.push_back(porig); }
    ctx->saved_data["m"] = std::move(meta);
    ctx->saved_data["d"] = std::vector<double>{lr, eps};
    ctx->saved_data["t"] = std::move(keep);
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grad_outputs) {
    const auto meta = ctx->saved_data["m"].toIntVector();
    const auto lre = ctx->saved_data["d"].toDoubleVector();
    const auto keep = ctx->saved_data["t"].toTensorVector();
    const int64_t optim = meta[0] & 255, det = (meta[0] >> 9) & 3, T = meta[1], nstate_own = meta[2], nstate = meta[2] + meta[3];
    const bool weighted = meta[4] != 0, psw_grad = meta[5] != 0, has_copt = meta[6] != 0, planned = meta[3] == 6;
    const int64_t nnz = meta[7];
    const std::vector<int64_t> p(meta.begin() + 8, meta.begin() + 8 + T);
    const std::vector<int64_t> q(meta.begin() + 8 + T, meta.begin() + 8 + 2 * T);
    const std::vector<int64_t> r(meta.begin() + 8 + 2 * T, meta.begin() + 8 + 3 * T + 1);
    const double lr = lre[0], eps = lre[1];
    Geom G;
    make_geom(G, 1, p, q, r);
    const ttx_geom& g = G.g;
    const Tensor& cache_weight = keep[0];
    const int64_t *tableidx_p, *pcol_p, *prow_p;
    const int32_t *ploc_p, *n_tt;  // n_tt: the device-side split point
    const void* plan_p;
    size_t at = 1;
    if (planned) {  // {tableidx, pcol, prow, ploc, n_tt, plan}
      tableidx_p = keep[1].data_ptr<int64_t>(); pcol_p = keep[2].data_ptr<int64_t>(); prow_p = keep[3].data_ptr<int64_t>();
      ploc_p = keep[4].data_ptr<int32_t>(); n_tt = keep[5].data_ptr<int32_t>(); plan_p = keep[6].data_ptr();
      at = 7;
    } else {
      const size_t ib = ((size_t)nnz * 8 + 255) / 256 * 256, lb = ((size_t)nnz * 4 + 255) / 256 * 256;
      const char* base = (const char*)keep[1].data_ptr();
      tableidx_p = (const int64_t*)base; pcol_p = (const int64_t*)(base + ib); prow_p = (const int64_t*)(base + 2 * ib);
      ploc_p = (const int32_t*)(base + 3 * ib); n_tt = (const int32_t*)(base + 3 * ib + lb); plan_p = base + 3 * ib + lb + 256;
      at = 2;
    }
    const Tensor cache_opt_state = has_copt ? keep[at++] : Tensor();
    const Tensor* cores = &keep[at];
    const Tensor* state = &keep[at + T];
    at += T + nstate_own;
    const Tensor ppsw = weighted ? keep[at++] : Tensor();
    const Tensor rows_keep_s = psw_grad ? keep[at] : Tensor(), porig_s = psw_grad ? keep[at + 1] : Tensor();

    variable_list grads(kHead + nstate + T);
    Tensor go = grad_outputs[0];
    TORCH_CHECK(go.defined(), "tt_embeddings: backward needs the output gradient");
    go = go.contiguous();
    TORCH_CHECK(go.scalar_type() == at::kFloat && go.dim() == 3 && go.size(0) == 1,
                "tt_embeddings: d_output must be float32 [1, B, D]");
    const int64_t B = go.size(1), D = go.size(2);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(go.device());
    auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();

    float* cp[TTX_MAX_CORES] = {};
    float* sp[TTX_MAX_CORES] = {};
    float* gp[TTX_MAX_CORES] = {};
    std::vector<Tensor> dense;
    for (int t = 0; t < T; ++t) {
      cp[t] = cores[t].data_ptr<float>();
      if (optim == TTX_OPTIM_ADAGRAD) sp[t] = state[t].data_ptr<float>();
      if (optim == TTX_OPTIM_DENSE) {
        dense.push_back(at::empty_like(cores[t]));
        gp[t] = dense.back().data_ptr<float>();
      }
    }
    const size_t wb = ttx_tt_backward_workspace_bytes(&g, (int32_t)B, (int32_t)D, nnz);
    Tensor ws = bytes_on(cache_weight, wb);
    int32_t scatter_done = 0;  // the cache rows' SGD scatter rode in the optimizer's launch (ttx_tt_backward_wc)
    // (profiles/r06_cache_bandwidth.md: from here on the sorted update is also the faster one on a Zipf
    //  stream -- near 300k lookups against the atomic SGD / dense scatter, near 60k against the atomic row-wise Adagrad)
    const int64_t kSortedAutoNnz = optim == TTX_OPTIM_ADAGRAD ? 65536 : 262144;
    const bool sorted = det == 1 || (det == 0 && nnz >= kSortedAutoNnz);
    if (optim == TTX_OPTIM_SGD && !ppsw.defined() && !sorted)
      check(ttx_tt_backward_wc(&g, (int32_t)optim, (int32_t)B, (int32_t)D, (float)lr, (float)eps, nnz, pcol_p, prow_p, tableidx_p,
                               nullptr, go.data_ptr<float>(), cp, nullptr, nullptr, plan_p, ws.data_ptr(), wb, stream,
                               n_tt, ploc_p, go.data_ptr<float>(), -(float)lr, cache_weight.data_ptr<float>(), &scatter_done));
    else
      check(ttx_tt_backward_w(&g, (int32_t)optim, (int32_t)B, (int32_t)D, (float)lr, (float)eps, nnz,
                              pcol_p, prow_p, tableidx_p,
                              ppsw.defined() ? ppsw.data_ptr<float>() : nullptr, go.data_ptr<float>(), cp,
                              optim == TTX_OPTIM_ADAGRAD ? sp : nullptr, optim == TTX_OPTIM_DENSE ? gp : nullptr,
                              plan_p, ws.data_ptr(), wb, stream));
    if (psw_grad) {  // gradient of the per_sample_weights (argument slot 13), in the caller's order
      const Tensor &rows_keep = rows_keep_s, &porig = porig_s;
      Tensor d_part = at::empty({nnz}, rows_keep.options());
      check(ttx_psw_backward((int32_t)B, (int32_t)D, nnz, rows_keep.data_ptr<float>(), prow_p, tableidx_p, go.data_ptr<float>(),
                             d_part.data_ptr<float>(), stream));
      Tensor d_psw = at::empty({nnz}, rows_keep.options());
      d_psw.index_copy_(0, porig.to(at::kLong), d_part);
      grads[13] = d_psw;
    }
    const int32_t* loc = ploc_p;
    const int64_t* rows = prow_p;
    const float* gcache = go.data_ptr<float>();  // (weighted: every cached lookup's own scaled gradient row)
    Tensor scaled, iota;
    if (ppsw.defined()) {
      scaled = at::empty({nnz, D}, go.options());
      iota = at::empty({nnz}, go.options().dtype(at::kLong));
      check(ttx_cache_weighted_grad_n(nnz, n_tt, (int32_t)D, go.data_ptr<float>(), rows, ppsw.data_ptr<float>(),
                                      scaled.data_ptr<float>(), iota.data_ptr<int64_t>(), stream));
      gcache = scaled.data_ptr<float>();
      rows = iota.data_ptr<int64_t>();
    }
    if (sorted) {  // atomic-free: the cached lookups grouped by cache row, one writer per row
      const int64_t nbags = ppsw.defined() ? nnz : B;  // (weighted: every lookup brings its own scaled gradient row)
      const size_t sb = ttx_cache_backward_sorted_workspace_bytes(nnz, nbags, (int32_t)D);
      Tensor sws = bytes_on(cache_weight, sb);
      Tensor gcw;
      float* dst = cache_weight.data_ptr<float>();
      if (optim == TTX_OPTIM_DENSE) {
        for (int t = 0; t < T; ++t) grads[kHead + nstate + t] = dense[t];
        gcw = at::empty_like(cache_weight);
        dst = gcw.data_ptr<float>();
        grads[12] = gcw;
      } else if (optim == TTX_OPTIM_ADAGRAD) {
        TORCH_CHECK(cache_opt_state.defined(), "tt_embeddings: Adagrad with a live cache needs cache_optimizer_state");
      }
      check(ttx_cache_backward_sorted((int32_t)optim, nnz, n_tt, nbags, (int32_t)D, gcache, loc, rows, (float)lr, (float)eps,
                                      cache_weight.size(0), optim == TTX_OPTIM_ADAGRAD ? cache_opt_state.data_ptr<float>() : nullptr,
                                      dst, sws.data_ptr(), sb, stream));
    } else if (optim == TTX_OPTIM_SGD) {
      if (!scatter_done)
        check(ttx_cache_backward_sgd_n(nnz, n_tt, (int32_t)D, gcache, loc, rows, (float)lr,
                                       cache_weight.data_ptr<float>(), stream));
    } else if (optim == TTX_OPTIM_ADAGRAD) {
      TORCH_CHECK(cache_opt_state.defined(), "tt_embeddings: Adagrad with a live cache needs cache_optimizer_state");
      check(ttx_cache_backward_rowwise_adagrad_approx_n(nnz, n_tt, (int32_t)D, gcache, loc, rows, (float)lr,
                                                        (float)eps, cache_opt_state.data_ptr<float>(),
                                                        cache_weight.data_ptr<float>(), stream));
    } else {
      for (int t = 0; t < T; ++t) grads[kHead + nstate + t] = dense[t];
      // (the reference returns no cache gradient when nothing was hit, tt_embeddings_ops.py:349-353;
      // with the split point on the device this node always returns one -- zeros in that case)
      Tensor gcw = at::empty_like(cache_weight);
      check(ttx_cache_backward_dense_n(nnz, n_tt, (int32_t)D, gcache, loc, rows, cache_weight.size(0),
                                       gcw.data_ptr<float>(), stream));
      grads[12] = gcw;  // cache_weight
    }
    return grads;
  }
};

Tensor lookup_cached(const Tensor& indices, const Tensor& offsets, std::vector<int64_t> p, std::vector<int64_t> q,
                     std::vector<int64_t> r, int64_t optim, double lr, double eps, const Tensor& hashtbl,
                     const Tensor& cache_freq, const Tensor& cache_state, c10::optional<Tensor> cache_opt_state,
                     const Tensor& cache_weight, std::vector<Tensor> state, std::vector<Tensor> cores,
                     std::vector<Tensor> pre, c10::optional<Tensor> per_sample_weights) {
  return TTCachedLookupOp::apply(indices, offsets, std::move(p), std::move(q), std::move(r), optim, lr, eps, hashtbl,
                                 cache_freq, cache_state, cache_opt_state, cache_weight, per_sample_weights,
                                 at::TensorList(pre), at::TensorList(state), at::TensorList(cores));
}

// The cache-live prologues of several batches at once (ttx_lookup_prologue_cached_multi: three launches per 16 batches):
// -> {tableidx, pcol, prow [nbatch, nnz] int64, ploc [nbatch, nnz] int32, n_tt [nbatch, 1] int32, plans [nbatch, stride]};
// the rows k of the six are what `lookup_cached(pre=)` takes for batch k.  Void after the next cache_populate.
std::vector<Tensor> prologue_cached_multi(std::vector<Tensor> indices, std::vector<Tensor> offsets, std::vector<int64_t> p,
                                          std::vector<int64_t> q, std::vector<int64_t> r, const Tensor& hashtbl,
                                          const Tensor& cache_freq, const Tensor& cache_state) {
  Geom G;
  make_geom(G, 1, p, q, r);
  const ttx_geom& g = G.g;
  const int64_t nbatch = (int64_t)indices.size();
  TORCH_CHECK(nbatch > 0 && offsets.size() == indices.size(), "tt_embeddings: one offsets tensor per indices tensor");
  const int64_t nnz = indices[0].numel(), nb = offsets[0].numel() - 1;
  std::vector<const int64_t*> ip(nbatch), op(nbatch);
  for (int64_t k = 0; k < nbatch; ++k) {
    TORCH_CHECK(indices[k].is_cuda() && indices[k].scalar_type() == at::kLong && indices[k].is_contiguous() &&
                    offsets[k].is_cuda() && offsets[k].scalar_type() == at::kLong && offsets[k].is_contiguous() &&
                    indices[k].numel() == nnz && offsets[k].numel() == nb + 1,
                "tt_embeddings: the batches of a multi-batch prologue must be contiguous int64 GPU tensors of one size");
    ip[k] = indices[k].data_ptr<int64_t>();
    op[k] = offsets[k].data_ptr<int64_t>();
  }
  TORCH_CHECK(nnz > 0 && nb > 0, "tt_embeddings: empty batch");
  const int64_t H = hashtbl.numel();
  TORCH_CHECK(H > 0 && cache_freq.numel() == H && cache_state.numel() == H && hashtbl.scalar_type() == at::kLong &&
                  cache_freq.scalar_type() == at::kLong && cache_state.scalar_type() == at::kInt,
              "tt_embeddings: hashtbl / cache_freq (int64) and cache_state (int32) must have hashtbl_size entries");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(indices[0].device());
  auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const auto lopt = indices[0].options(), iopt = indices[0].options().dtype(at::kInt);
  Tensor rowidx = at::empty({nbatch, nnz}, lopt), tableidx = at::empty({nbatch, nnz}, lopt);
  Tensor pcol = at::empty({nbatch, nnz}, lopt), prow = at::empty({nbatch, nnz}, lopt);
  Tensor ploc = at::empty({nbatch, nnz}, iopt), n_tt = at::empty({nbatch, 1}, iopt);
  const size_t stride = (ttx_plan_bytes(&g, nnz) + 255) / 256 * 256;
  Tensor plans = at::empty({nbatch, (int64_t)stride}, lopt.dtype(at::kByte));
  const size_t wb = ttx_lookup_prologue_cached_multi_workspace_bytes((int32_t)nbatch, nnz);
  Tensor ws = bytes_on(indices[0], wb);
  check(ttx_lookup_prologue_cached_multi(&g, (int32_t)nbatch, nnz, ip.data(), nb, op.data(), H, hashtbl.data_ptr<int64_t>(),
                                         cache_freq.data_ptr<int64_t>(), cache_state.data_ptr<int32_t>(),
                                         rowidx.data_ptr<int64_t>(), tableidx.data_ptr<int64_t>(), pcol.data_ptr<int64_t>(),
                                         prow.data_ptr<int64_t>(), ploc.data_ptr<int32_t>(), n_tt.data_ptr<int32_t>(),
                                         plans.data_ptr(), stride, ws.data_ptr(), wb, stream));
  return {tableidx, pcol, prow, ploc, n_tt, plans};
}

// ---- direct RCCL exchange (table-sharded multi-GPU lookup, ttx_sharded.py) -----------------------------
// torch.distributed runs its collectives on a side stream of its own: every all_to_all costs two event hops
// and ~30 us of wrapper time, and its watchdog aborts when a collective is captured into a hipGraph
// (DESIGN.md section 7).  These four calls talk to the RCCL library torch already loaded, on the CURRENT
// stream: equal-split all-to-all (ncclAllToAll) and per-peer counts (a ncclSend/ncclRecv group), both capturable.  Types, enums and
// prototypes are <rccl/rccl.h>'s (the ROCm header; the library is the one torch loaded -- rccl_abi_check() below refuses a library
// whose major version differs from the header's).
static void rccl_abi_check() {
  static const bool ok = [] {
    int v = 0;
    TORCH_CHECK(ncclGetVersion(&v) == ncclSuccess, "RCCL ncclGetVersion failed");
    // (version code: major * 10000 + minor * 100 + patch since 2.9)
    TORCH_CHECK(v / 10000 == NCCL_MAJOR, "RCCL library version ", v, " does not match the header this file was built with (",
                NCCL_VERSION_CODE, ")");
    return true;
  }();
  (void)ok;
}

void rccl_check(ncclResult_t rc, const char* what) { TORCH_CHECK(rc == ncclSuccess, "RCCL ", what, ": ", ncclGetErrorString(rc)); }

pybind11::bytes rccl_unique_id() {
  rccl_abi_check();
  ncclUniqueId id;
  rccl_check(ncclGetUniqueId(&id), "ncclGetUniqueId");
  return pybind11::bytes(id.internal, sizeof(id.internal));
}

int64_t rccl_comm_init(const std::string& id_bytes, int64_t rank, int64_t world, int64_t device_index) {
  rccl_abi_check();
  TORCH_CHECK(id_bytes.size() == sizeof(ncclUniqueId), "bad RCCL unique id");
  ncclUniqueId id;
  memcpy(id.internal, id_bytes.data(), sizeof(id.internal));
  ncclComm_t comm = nullptr;
  {
    pybind11::gil_scoped_release nogil;  // collective call: blocks until every rank has arrived
    c10::hip::HIPGuardMasqueradingAsCUDA guard(c10::Device(c10::kCUDA, (c10::DeviceIndex)device_index));
    rccl_check(ncclCommInitRank(&comm, (int)world, id, (int)rank), "ncclCommInitRank");
  }
  return (int64_t)(intptr_t)comm;
}

void rccl_comm_destroy(int64_t comm) {
  if (comm) (void)ncclCommDestroy((ncclComm_t)(intptr_t)comm);
}

// out[p] <- block p of rank p's `in`, blocks of in.numel() / world elements; on the current stream
void rccl_all_to_all(int64_t comm, const Tensor& out, const Tensor& in, int64_t world) {
  TORCH_CHECK(out.is_cuda() && in.is_cuda() && out.is_contiguous() && in.is_contiguous() &&
                  out.scalar_type() == in.scalar_type() && out.numel() == in.numel() && in.numel() % world == 0,
              "rccl_all_to_all: contiguous GPU tensors of one dtype and size, divisible by the world size");
  ncclDataType_t dt;
  switch (in.scalar_type()) {
    case at::kFloat: dt = ncclFloat32; break;
    case at::kLong: dt = ncclInt64; break;
    case at::kInt: dt = ncclInt32; break;
    default: TORCH_CHECK(false, "rccl_all_to_all: float32 / int64 / int32 only");
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(in.device());
  auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  rccl_check(ncclAllToAll(in.data_ptr(), out.data_ptr(), (size_t)(in.numel() / world), dt,
                          (ncclComm_t)(intptr_t)comm, stream), "ncclAllToAll");
}

// Uneven splits (26 tables on 8 ranks: 4,4,3,3,3,3,3,3 table blocks per owner): block p of `in` holds
// send_counts[p] elements for rank p, block p of `out` receives recv_counts[p] elements from rank p.  One
// ncclSend + ncclRecv per peer inside a group -- over xGMI every pair of GPUs has its own link, so the group is
// W - 1 concurrent point-to-point transfers -- on the current stream, capturable like ncclAllToAll.
void rccl_all_to_allv(int64_t comm, const Tensor& out, const Tensor& in, const std::vector<int64_t>& send_counts,
                      const std::vector<int64_t>& recv_counts) {
  TORCH_CHECK(out.is_cuda() && in.is_cuda() && out.is_contiguous() && in.is_contiguous() &&
                  out.scalar_type() == in.scalar_type() && send_counts.size() == recv_counts.size(),
              "rccl_all_to_allv: contiguous GPU tensors of one dtype, one count per rank and direction");
  ncclDataType_t dt;
  switch (in.scalar_type()) {
    case at::kFloat: dt = ncclFloat32; break;
    case at::kLong: dt = ncclInt64; break;
    case at::kInt: dt = ncclInt32; break;
    default: TORCH_CHECK(false, "rccl_all_to_allv: float32 / int64 / int32 only");
  }
  int64_t ns = 0, nr = 0;
  for (size_t p = 0; p < send_counts.size(); ++p) {
    TORCH_CHECK(send_counts[p] >= 0 && recv_counts[p] >= 0, "rccl_all_to_allv: negative count");
    ns += send_counts[p];
    nr += recv_counts[p];
  }
  TORCH_CHECK(ns == in.numel() && nr == out.numel(), "rccl_all_to_allv: counts do not add up to the tensor sizes");
  const size_t esz = (size_t)in.element_size();
  c10::hip::HIPGuardMasqueradingAsCUDA guard(in.device());
  auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  ncclComm_t c = (ncclComm_t)(intptr_t)comm;
  rccl_check(ncclGroupStart(), "ncclGroupStart");
  size_t so = 0, ro = 0;
  ncclResult_t rc = ncclSuccess;
  for (size_t p = 0; p < send_counts.size() && rc == ncclSuccess; ++p) {
    if (send_counts[p]) rc = ncclSend((const char*)in.data_ptr() + so * esz, (size_t)send_counts[p], dt, (int)p, c, stream);
    if (rc == ncclSuccess && recv_counts[p]) rc = ncclRecv((char*)out.data_ptr() + ro * esz, (size_t)recv_counts[p], dt, (int)p, c, stream);
    so += (size_t)send_counts[p];
    ro += (size_t)recv_counts[p];
  }
  const ncclResult_t rc_end = ncclGroupEnd();  // (always close the group)
  rccl_check(rc, "ncclSend/ncclRecv");
  rccl_check(rc_end, "ncclGroupEnd");
}

// ---- out.backward(grad) of a lookup's OWN output, past autograd's engine ---------------------------------------------------
// The reference benchmark's loop is `tt_emb(indices, offsets).backward(grad)` per request (tt_embeddings_benchmark.py:94-108).
// With a fused optimizer the graph under that output is ONE node (this file's) whose backward returns no gradient to anybody:
// the engine's graph task, its hand-over to the device thread and back cost ~40 us of host time per step (scripts/host_time.py) --
// more than the step's kernels take -- to call one function.  NodeRef::backward calls it on the calling thread.  Only where that
// is exactly what the engine would do: the module installs it for fused optimizers without a weight gradient only, and every
// condition below that the engine would treat differently (hooks, anomaly mode, another stream, a gradient that is itself part
// of a graph, a wrong shape) returns false -> the caller takes Tensor.backward's ordinary route.
struct NodeRef {
  std::shared_ptr<torch::autograd::Node> fn;
  int64_t num_tables = 0, B = 0, D = 0;
  bool backward(const Tensor& grad) {
    using torch::autograd::Node;
    Node* n = fn.get();
    if (!n || !grad.defined() || !grad.is_cuda() || grad.scalar_type() != at::kFloat || grad.requires_grad()) return false;
    if (!n->tensor_pre_hooks().empty() || !n->pre_hooks().empty() || !n->post_hooks().empty() ||
        !n->retains_grad_hooks().empty() || torch::autograd::AnomalyMode::is_enabled())
      return false;
    Tensor g = grad;
    if (g.dim() == 2 && num_tables == 1) g = g.unsqueeze(0);  // (TTEmbeddingBag hands out the squeezed view)
    if (g.dim() != 3 || g.size(0) != num_tables || g.size(1) != B || g.size(2) != D) return false;
    // the engine runs a node on the stream its forward ran on; only the same stream is the same thing here
    const auto fs = n->stream();
    if (fs.has_value() && (fs->device_index() != g.get_device() ||
                           *fs != c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(fs->device_index()).unwrap()))
      return false;
    at::AutoGradMode no_grad(false);
    variable_list outs = (*n)(variable_list{g});
    for (const auto& o : outs) TORCH_INTERNAL_ASSERT(!o.defined(), "tt_embeddings: the direct backward is for fused optimizers only");
    n->release_variables();  // (as the engine does without retain_graph)
    return true;
  }
  // `out.backward(grad)` for the registered tensor `out`: additionally requires that out's grad_fn is STILL this node (or, for the
  // squeezed view TTEmbeddingBag hands out, a SqueezeBackward whose only input edge is this node).  An in-place op on the output
  // (`out.mul_(2)`, `out += bias`) keeps the tensor's identity but rebases its grad_fn: that is no longer the plain case, and the
  // engine must run the in-place op's backward first.
  bool backward_of(const Tensor& out, const Tensor& grad) {
    using torch::autograd::Node;
    const std::shared_ptr<Node>& gf = out.grad_fn();
    if (!gf || !fn) return false;
    if (gf.get() != fn.get()) {
      const std::string nm = gf->name();
      if (nm.rfind("SqueezeBackward", 0) != 0 || gf->num_outputs() != 1 || gf->next_edge(0).function.get() != fn.get() ||
          !gf->tensor_pre_hooks().empty() || !gf->pre_hooks().empty() || !gf->post_hooks().empty() || !gf->retains_grad_hooks().empty())
        return false;
    }
    return backward(grad);
  }
  // the node as Python sees it (tensor.grad_fn): a root for torch.autograd.backward when the tensor itself is gone
  pybind11::object node() const { return pybind11::reinterpret_steal<pybind11::object>(torch::autograd::functionToPyObject(fn)); }
};
NodeRef node_of(const Tensor& out) {
  NodeRef r;
  r.fn = out.grad_fn();
  TORCH_CHECK(r.fn && out.dim() == 3 &&
                  (dynamic_cast<torch::autograd::CppNode<TTLookupOp>*>(r.fn.get()) != nullptr ||
                   dynamic_cast<torch::autograd::CppNode<TTCachedLookupOp>*>(r.fn.get()) != nullptr),
              "tt_embeddings: node_of() takes the tensor lookup() / lookup_cached() returned");
  r.num_tables = out.size(0);
  r.B = out.size(1);
  r.D = out.size(2);
  return r;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  pybind11::class_<NodeRef>(m, "NodeRef")
      .def("backward", &NodeRef::backward,
           "run the node's backward (fused optimizer step) on the calling thread; False = not the plain case, use autograd")
      .def("backward_of", &NodeRef::backward_of,
           "backward(grad) after checking that the given tensor's grad_fn still is this node (or a squeeze view of it)")
      .def("node", &NodeRef::node, "the node as tensor.grad_fn would return it");
  m.def("node_of", &node_of, "the autograd node behind an output of lookup() / lookup_cached()");
  m.doc() = "native autograd node of the TT lookup (cache not live) over the C ABI of libttx.so";
  m.def("lookup", &lookup, "prologue + forward; backward = fused optimizer step or dense core gradients",
        pybind11::arg("indices"), pybind11::arg("offsets"), pybind11::arg("num_tables"), pybind11::arg("p"),
        pybind11::arg("q"), pybind11::arg("r"), pybind11::arg("optim"), pybind11::arg("lr"), pybind11::arg("eps"),
        pybind11::arg("hashtbl"), pybind11::arg("cache_freq"), pybind11::arg("state"), pybind11::arg("cores"),
        pybind11::arg("per_sample_weights") = pybind11::none(), pybind11::arg("pre_rowidx") = pybind11::none(),
        pybind11::arg("pre_tableidx") = pybind11::none(), pybind11::arg("pre_plan") = pybind11::none());
  m.def("prologue", &prologue, "the lookup prologue alone, on the current stream: -> [rowidx, tableidx, plan] for lookup(pre_*=)",
        pybind11::arg("indices"), pybind11::arg("offsets"), pybind11::arg("num_tables"), pybind11::arg("p"),
        pybind11::arg("q"), pybind11::arg("r"), pybind11::arg("hashtbl") = pybind11::none(),
        pybind11::arg("cache_freq") = pybind11::none(), pybind11::arg("n_dev") = pybind11::none());
  m.def("prologue_multi", &prologue_multi, "the prologues of several equal-sized batches in one launch: -> [rowidx, tableidx, plans], row k for batch k");
  m.def("lookup_cached", &lookup_cached, "cache-live lookup of one table: partition, contraction of the misses, gather of the hits",
        pybind11::arg("indices"), pybind11::arg("offsets"), pybind11::arg("p"), pybind11::arg("q"), pybind11::arg("r"),
        pybind11::arg("optim"), pybind11::arg("lr"), pybind11::arg("eps"), pybind11::arg("hashtbl"),
        pybind11::arg("cache_freq"), pybind11::arg("cache_state"), pybind11::arg("cache_optimizer_state"),
        pybind11::arg("cache_weight"), pybind11::arg("state"), pybind11::arg("cores"),
        pybind11::arg("pre") = std::vector<Tensor>(), pybind11::arg("per_sample_weights") = pybind11::none());
  m.def("prologue_cached_multi", &prologue_cached_multi,
        "the cache-live prologues of several equal-sized batches: -> [tableidx, pcol, prow, ploc, n_tt, plans], row k for batch k");
  m.def("rccl_unique_id", &rccl_unique_id, "ncclGetUniqueId (rank 0; broadcast the bytes to the others)");
  m.def("rccl_comm_init", &rccl_comm_init, "ncclCommInitRank on the given device (collective; releases the GIL)");
  m.def("rccl_comm_destroy", &rccl_comm_destroy, "ncclCommDestroy (collective; releases the GIL: callers run it under a timeout)",
        pybind11::call_guard<pybind11::gil_scoped_release>());
  m.def("rccl_all_to_all", &rccl_all_to_all, "equal-split all-to-all on the current stream (capturable)");
  m.def("rccl_all_to_allv", &rccl_all_to_allv, "all-to-all with per-peer element counts (ncclSend/ncclRecv group) on the current stream (capturable)");
  m.def("abi_version", []() { return ttx_version(); });
}
