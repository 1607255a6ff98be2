"""Oracle for the sufficient-statistics form of ``EigenFunctionTask.loss_func`` (core.py:387-457).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  The reference evaluates the loss on one
batch in one process; the data-parallel build reduces every rank's slice to the vector of batch sums
below, adds the vectors across ranks, and evaluates the (rational) scalar tail on the sum
(SURVEY.md section 8e).  These functions restate that factorisation in plain PyTorch so that

* ``tests/test_oracle_golden.py`` can check  loss_from_stats(batch_stats(.)) == reference loss, and
* ``tests/test_dist_gloo.py`` can check, over a real 2-process gloo group, that sharding + two
  all-reduces reproduce the single-process loss and parameter gradient.

Layout of the vector (same as ``include/cvf.h``):
  generator: [W, S1(k), S2(i<=j, row-major), E(k)]           E_i = sum_b w_b sum_j a_j (d y_bi / d x_bj)^2
  transfer : [W, S1(k), S2(i<=j), W', S1'(k), S2'_ii(k), T(k)]   T_i = sum_b w_b (y'_bi - y_bi)^2
"""

import numpy as np
import torch


def batch_stats(y, w, dirichlet=None, y_lag=None, w_lag=None):
    k = y.shape[1]
    parts = [w.sum().reshape(1), (y * w[:, None]).sum(0)]
    parts.append(torch.stack([(y[:, i] * y[:, j] * w).sum() for i in range(k) for j in range(i, k)]))
    if y_lag is None:
        parts.append((dirichlet * w[:, None]).sum(0))                     # dirichlet [B,k]: sum_j a_j G_bji^2
    else:
        parts += [w_lag.sum().reshape(1), (y_lag * w_lag[:, None]).sum(0), (y_lag ** 2 * w_lag[:, None]).sum(0),
                  ((y_lag - y) ** 2 * w[:, None]).sum(0)]
    return torch.cat(parts)


def loss_from_stats(s, k, *, alpha, eig_w, beta=1.0, lag_idx=0, dt=1.0, sort_eigvals=True):
    """Returns (loss, eig_sorted, npl, pen, cvec); differentiable w.r.t. ``s`` with eig / cvec constant."""
    npair = k * (k + 1) // 2
    W = s[0]
    S1 = s[1:1 + k]
    S2 = {}
    p = 1 + k
    for i in range(k):
        for j in range(i, k):
            S2[(i, j)] = s[p]
            p += 1
    o = 1 + k + npair
    m = [S1[i] / W for i in range(k)]                                     # core.py:409
    v = [S2[(i, i)] / W - m[i] ** 2 for i in range(k)]                    # core.py:410
    if lag_idx == 0:
        num = [s[o + i] for i in range(k)]
        den = v
        pref = 1.0 / (W * beta)                                           # core.py:426,438
    else:
        Wl = s[o]
        ml = [s[o + 1 + i] / Wl for i in range(k)]
        vl = [s[o + 1 + k + i] / Wl - ml[i] ** 2 for i in range(k)]       # core.py:416
        num = [s[o + 1 + 2 * k + i] for i in range(k)]
        den = [v[i] + vl[i] for i in range(k)]
        pref = 1.0 / (dt * lag_idx) / W                                   # core.py:428,440
    eig = torch.stack([(pref * num[i] / den[i]).detach() for i in range(k)])
    cvec = np.argsort(eig.numpy(), kind="stable") if sort_eigvals else np.arange(k)
    npl = 0.0
    for idx in range(k):
        c = int(cvec[idx])
        src = c if lag_idx == 0 else idx                                  # core.py:438 vs :440
        npl = npl + eig_w[idx] * num[src] / den[c]
    npl = pref * npl
    pen = sum((v[i] - 1.0) ** 2 for i in range(k))                        # core.py:446
    for i in range(k):
        for j in range(i + 1, k):
            pen = pen + (S2[(i, j)] / W - m[i] * m[j]) ** 2               # core.py:452
    return npl + alpha * pen, eig[torch.as_tensor(cvec)], npl, pen, cvec
