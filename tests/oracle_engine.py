"""A stand-in for the native module `tt_embeddings` backed by the CPU oracle.

TEST INFRASTRUCTURE ONLY.  Tests install it with
`monkeypatch.setattr(tt_embeddings_ops, "_engine", oracle_engine)` to drive the
Python module surface (autograd plumbing, cache life-cycle, state_dict) on CPU
tensors in the GPU-less build container -- the same way the reference's Python
can be driven on top of a stub `tt_embeddings`.  The product
never imports this file.
"""
import numpy as np
import torch

import oracle_lib as O


def _np(t):
    return t.detach().cpu().numpy()


def _geom(num_tables, p, q, r):
    return O.make_geom(num_tables, p, q, r)


def tt_forward(batch_count, num_tables, B, D, p, q, r, L, nnz, indices, rowidx, tableidx, tt_cores):
    out = O.tt_forward(_geom(num_tables, p, q, r), B, D, _np(indices), _np(rowidx), _np(tableidx),
                       [_np(c) for c in tt_cores], nnz=nnz)
    return torch.from_numpy(out)


def tt_dense_backward(batch_count, D, p, q, r, L, nnz, indices, rowidx, tableidx, d_output, tt_cores):
    nt = tt_cores[0].size(0)
    grads = O.tt_backward(_geom(nt, p, q, r), O.OPTIM_DENSE, d_output.size(1), D, 0.0, 0.0, _np(indices), _np(rowidx),
                          _np(tableidx), _np(d_output), [np.ascontiguousarray(_np(c)) for c in tt_cores], nnz=nnz)
    return [torch.from_numpy(g) for g in grads]


def _inplace(optim, D, lr, eps, p, q, r, nnz, indices, rowidx, tableidx, d_output, tt_cores, state):
    nt = tt_cores[0].size(0)
    cores = [np.ascontiguousarray(_np(c)).copy() for c in tt_cores]
    st = [np.ascontiguousarray(_np(s)).copy() for s in state] if state is not None else None
    O.tt_backward(_geom(nt, p, q, r), optim, d_output.size(1), D, lr, eps, _np(indices), _np(rowidx), _np(tableidx),
                  _np(d_output), cores, st, nnz=nnz)
    with torch.no_grad():
        for c, n in zip(tt_cores, cores):
            c.copy_(torch.from_numpy(n))
        if st is not None:
            for s, n in zip(state, st):
                s.copy_(torch.from_numpy(n))


def tt_sgd_backward(batch_count, D, lr, p, q, r, L, nnz, indices, rowidx, tableidx, d_output, tt_cores):
    _inplace(O.OPTIM_SGD, D, lr, 0.0, p, q, r, nnz, indices, rowidx, tableidx, d_output, tt_cores, None)


def tt_adagrad_backward(batch_count, D, lr, eps, p, q, r, L, nnz, indices, rowidx, tableidx, d_output, optimizer_state, tt_cores):
    _inplace(O.OPTIM_ADAGRAD, D, lr, eps, p, q, r, nnz, indices, rowidx, tableidx, d_output, tt_cores, list(optimizer_state))


def update_cache_state(indices, hashtbl, cache_freq):
    h, f = _np(hashtbl), _np(cache_freq)  # share memory with the CPU tensors
    O.update_cache_state(_np(indices), h, f)


def cache_populate(num_embeddings, p, q, r, tt_cores, L, hashtbl, cache_freq, cache_state, cache_weight, reference_exact=False):
    cw = _np(cache_weight)
    O.cache_populate(_geom(tt_cores[0].size(0), p, q, r), [_np(c) for c in tt_cores], _np(hashtbl), _np(cache_freq),
                     _np(cache_state), cw)


def preprocess_indices_sync(colidx, offsets, num_tables, warmup, hashtbl, cache_state):
    if colidx.numel() == 0:
        e = torch.empty_like(colidx)
        return colidx, e, e.clone(), 0, None
    c, r, t, n, loc = O.preprocess_indices(_np(colidx), _np(offsets), num_tables, warmup, _np(hashtbl), _np(cache_state))
    return (torch.from_numpy(np.ascontiguousarray(c)), torch.from_numpy(r), torch.from_numpy(t), n,
            None if loc is None else torch.from_numpy(loc))


def cache_forward(B, nnz, cache_locations, rowidx, cache_weight, output):
    O.cache_forward(B, _np(cache_locations)[:nnz], _np(rowidx)[:nnz], _np(cache_weight), _np(output)[0])


def cache_backward_sgd(nnz, grad_output, cache_locations, rowidx, lr, cache_weight):
    O.cache_backward_sgd(_np(grad_output).reshape(-1, cache_weight.size(1)), _np(cache_locations)[:nnz], _np(rowidx)[:nnz],
                         lr, _np(cache_weight))


def cache_backward_dense(nnz, grad_output, cache_locations, rowidx, lr, cache_weight):
    D = cache_weight.size(1)
    return torch.from_numpy(O.cache_backward_dense(_np(grad_output).reshape(-1, D), _np(cache_locations)[:nnz],
                                                   _np(rowidx)[:nnz], cache_weight.size(0), D))


def cache_backward_rowwise_adagrad_approx(nnz, grad_output, cache_locations, rowidx, lr, eps, cache_optimizer_state, cache_weight):
    O.cache_backward_rowwise_adagrad_approx(_np(grad_output).reshape(-1, cache_weight.size(1)), _np(cache_locations)[:nnz],
                                            _np(rowidx)[:nnz], lr, eps, _np(cache_optimizer_state).reshape(-1), _np(cache_weight))
