"""CPU: the sufficient-statistics factorisation (oracle/stats.py) equals the reference's loss_func
(through the golden fixtures), value AND parameter gradient - the identity the data-parallel path rests on."""

import numpy as np
import pytest
import torch

from oracle import nnref, stats
from tests import goldens
from tests.test_oracle_golden import build_pp


@pytest.fixture(autouse=True)
def _f64():
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(torch.float32)


def local_stats(sd, k, pp, X, w, a, Xl, wl):
    y = nnref.eigenfunctions_forward(sd, k, pp(X))
    if Xl is None:
        tot = X[0].numel()
        G = torch.stack([torch.autograd.grad(y[:, i].sum(), X, create_graph=True)[0].reshape(-1, tot) for i in range(k)], dim=2)
        return stats.batch_stats(y, w, dirichlet=(G ** 2 * a[None, :, None]).sum(1))
    return stats.batch_stats(y, w, y_lag=nnref.eigenfunctions_forward(sd, k, pp(Xl)), w_lag=wl)


@pytest.mark.parametrize("name", goldens.KAT_CASES)
def test_loss_from_stats_matches_reference(name):
    g = goldens.load(name, "f64")
    k, lag = int(g["k"]), int(g["lag_idx"])
    sd = {n: p.requires_grad_(True) for n, p in goldens.state_dict(g, dtype=torch.float64).items()}
    traj, w = np.array(g["traj"]), np.array(g["w"])
    B = traj.shape[0] - lag
    X = torch.tensor(traj[:B], dtype=torch.float64, requires_grad=(lag == 0))
    Xl = torch.tensor(traj[lag:lag + B], dtype=torch.float64) if lag else None
    wl = torch.tensor(w[lag:lag + B]) if lag else None
    a = torch.tensor(np.array(g["diag_coeff"])) if lag == 0 else None
    s = local_stats(sd, k, build_pp(g), X, torch.tensor(w[:B]), a, Xl, wl)
    loss, eig, npl, pen, cvec = stats.loss_from_stats(s, k, alpha=float(g["alpha"]), eig_w=list(g["eig_w"]), beta=float(g["beta"]),
                                                      lag_idx=lag, dt=float(g["dt"]), sort_eigvals=bool(g["sort"]))
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-10)
    np.testing.assert_allclose(eig.numpy(), g["eig"], rtol=1e-10)
    assert list(cvec) == list(g["cvec"])
    for n, p in sd.items():
        np.testing.assert_allclose(p.grad.numpy(), g["grad/" + n], rtol=1e-8, atol=1e-10, err_msg=n)
