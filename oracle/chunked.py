"""Oracle for ``EigenFunctionTask.loss_func`` + ``backward`` (core.py:387-457, 517) on batches too large for one autograd graph.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  The loss is a rational function of batch sums (``oracle/stats.py``,
pinned against the reference's fixtures by ``tests/test_oracle_stats.py``), so a batch is walked in chunks, twice:

  pass 1   s = sum over chunks of batch_stats(chunk)                       (no parameter graph kept)
           loss, eig, npl, pen, cvec = loss_from_stats(s);   c = d loss / d s
  pass 2   d loss / d theta = sum over chunks of d (c . batch_stats(chunk)) / d theta

which is the reference's value and gradient up to fp64 summation order (checked against ``losses.ef_loss`` on batches that
fit, ``tests/test_oracle_stats.py``).  Peak memory is one chunk's double-backward graph.
"""

import torch

from . import nnref, stats


def _chunk_stats(sd, k, pp, X, w, a, X_lag, w_lag, create_graph, activation):
    y = nnref.eigenfunctions_forward(sd, k, pp(X), activation)
    if X_lag is not None:
        return stats.batch_stats(y, w, y_lag=nnref.eigenfunctions_forward(sd, k, pp(X_lag), activation), w_lag=w_lag)
    B = X.shape[0]
    G = [torch.autograd.grad(y[:, i].sum(), X, retain_graph=True, create_graph=create_graph)[0].reshape(B, -1) for i in range(k)]
    return stats.batch_stats(y, w, dirichlet=torch.stack([(g ** 2 * a).sum(dim=1) for g in G], dim=1))


def ef_loss_and_grad(sd, k, pp, X, w, X_lag=None, w_lag=None, *, alpha, eig_w, diag_coeff=None, beta=1.0, lag_idx=0, dt=1.0,
                     sort_eigvals=True, activation=torch.tanh, chunk=64):
    """``sd``: name -> parameter tensor (leaf, requires_grad).  Returns ((loss, eig, npl, pen, cvec), {name: grad})."""
    names = list(sd.keys())
    params = [sd[n] for n in names]
    dtype = params[0].dtype
    B = X.shape[0]
    a = None
    if lag_idx == 0:
        a = diag_coeff.to(dtype) if diag_coeff is not None else torch.ones(X[0].numel(), dtype=dtype)

    def pieces(create_graph):
        for s0 in range(0, B, chunk):
            Xc = X[s0:s0 + chunk].detach().to(dtype)
            if lag_idx == 0:
                Xc.requires_grad_(True)
            Xl = None if X_lag is None else X_lag[s0:s0 + chunk].detach().to(dtype)
            wl = None if w_lag is None else w_lag[s0:s0 + chunk].to(dtype)
            yield _chunk_stats(sd, k, pp, Xc, w[s0:s0 + chunk].to(dtype), a, Xl, wl, create_graph, activation)

    total = None
    for s in pieces(False):
        total = s.detach() if total is None else total + s.detach()
    total.requires_grad_(True)
    loss, eig, npl, pen, cvec = stats.loss_from_stats(total, k, alpha=alpha, eig_w=eig_w, beta=beta, lag_idx=lag_idx, dt=dt,
                                                       sort_eigvals=sort_eigvals)
    (coef,) = torch.autograd.grad(loss, total)
    grads = [torch.zeros_like(p) for p in params]
    for s in pieces(True):
        for acc, g in zip(grads, torch.autograd.grad((coef * s).sum(), params, allow_unused=True)):
            if g is not None:
                acc += g
    return (loss.detach(), eig, npl.detach(), pen.detach(), cvec), dict(zip(names, grads))
