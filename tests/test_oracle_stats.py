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


def test_chunked_two_pass_oracle_equals_the_plain_oracle():
    """oracle/chunked.py (the batch walked in chunks, twice: sums, then the gradient with the global coefficients) against
    oracle/losses.ef_loss on a batch that fits one autograd graph - generator mode through the alignment layer with mixed
    features (grouped evaluation, oracle.pp compact=True) and transfer mode; loss, eigenvalues, ordering, every gradient."""
    from oracle import chunked, losses
    from oracle.pp import AlignFeature
    from tests.synth import make_molecule_traj, diag_coeff_for
    torch.set_default_dtype(torch.float64)
    try:
        n_atoms, B, k = 40, 150, 3
        traj, w, ref = make_molecule_traj(n_atoms, B + 2, seed=91, dtype=np.float64)
        rs = np.random.RandomState(4)
        feats = [("position", (0, 3, 7, 9))] + [("dihedral", tuple(int(i) for i in rs.choice(n_atoms, 4, replace=False))) for _ in range(5)]
        feats += [("bond", (1, 2)), ("angle", (4, 5, 6)), ("position", (11, 12)), ("bond", (20, 30)), ("angle", (9, 8, 7))]
        align = list(range(0, n_atoms, 2))
        for angle_value in (False, True):
            plain = AlignFeature(align, ref[align], feats, angle_value)
            grouped = AlignFeature(align, ref[align], feats, angle_value, compact=True)
            Xp = torch.tensor(traj[:7])
            np.testing.assert_allclose(grouped(Xp).numpy(), plain(Xp).numpy(), rtol=1e-13, atol=1e-14)
        d_r = plain(Xp).shape[1]
        sd0 = nnref.init_eigenfunctions([d_r, 12, 12, 1], k, torch.Generator().manual_seed(2), torch.float64)
        a = torch.tensor(diag_coeff_for(n_atoms, 3))
        for lag in (0, 2):
            kw = dict(alpha=15.0, eig_w=[1.0, 0.6, 0.3], diag_coeff=a if lag == 0 else None, beta=1.3, lag_idx=lag, dt=0.5)
            X, W = torch.tensor(traj[:B]), torch.tensor(w[:B])
            Xl, Wl = (torch.tensor(traj[lag:lag + B]), torch.tensor(w[lag:lag + B])) if lag else (None, None)
            sd = {n: p.clone().requires_grad_(True) for n, p in sd0.items()}
            (l1, e1, n1, p1, c1), gr = chunked.ef_loss_and_grad(sd, k, grouped, X, W, Xl, Wl, chunk=32, **kw)
            sd2 = {n: p.clone().requires_grad_(True) for n, p in sd0.items()}
            Xg = X.clone().requires_grad_(True) if lag == 0 else X
            l2, e2, n2, p2, c2 = losses.ef_loss(sd2, k, plain, Xg, W, Xl, Wl, **kw)
            l2.backward()
            np.testing.assert_allclose([float(l1), float(n1), float(p1)], [float(l2.detach()), float(n2.detach()), float(p2.detach())], rtol=1e-11)
            np.testing.assert_allclose(e1.numpy(), e2.numpy(), rtol=1e-11)
            assert list(c1) == list(c2)
            gmax = max(float(p.grad.abs().max()) for p in sd2.values())
            for n in sd2:
                np.testing.assert_allclose(gr[n].numpy(), sd2[n].grad.numpy(), rtol=0, atol=1e-11 * gmax, err_msg=n)
    finally:
        torch.set_default_dtype(torch.float32)
